// pvw_decrypt.hip -- decrypt_party_value's inner products on gfx950 (decryption.rs:257-274): HBM-bound streams over the
// ciphertexts in the layout they arrive in, [dealer][j][limb][slot]; three launch shapes chosen by the polynomial size.
#include <hip/hip_runtime.h>

#include "pvw_arith.h"
#include "pvw_chacha.h"
#include "pvw_decode.h"
#include "pvw_kernels.h"
#include "pvw_dev.h"

namespace pvw {

// ------------------------------------------------------------------------------------
// decrypt_mac: noisy[d] = sum_j shat[j] (.) c1s[d][j] - c2col[d]   (decryption.rs:257-274)
// on the ciphertext layout as it arrives, [d][j][L][l].  One workgroup per dealer; thread
// (g, e) owns slot pair e of the polynomial and the j = g, g+c, g+2c, ... terms.
// ------------------------------------------------------------------------------------
template <int U, bool DBUF>
__global__ __launch_bounds__(1024) void decrypt_mac_kernel(const u64* __restrict__ c1s,
                                                            const u64* __restrict__ shat,
                                                            const u64* __restrict__ c2col,
                                                            u64* __restrict__ noisy,
                                                            const Mod* __restrict__ mods, u32 k,
                                                            u32 ell, u32 pairs, u32 c,
                                                            u32 pair0_step) {
  extern __shared__ v2u64 dl[];
  const u32 d = blockIdx.x;
  const u32 pair_base = blockIdx.y * pair0_step;
  const u32 chunk = (pairs - pair_base) < pair0_step ? (pairs - pair_base) : pair0_step;
  const u32 g = threadIdx.x / chunk, el = threadIdx.x % chunk;
  const bool active = threadIdx.x < c * chunk;
  const u32 e = pair_base + el;
  const v2u64* cp = reinterpret_cast<const v2u64*>(c1s) + (size_t)d * k * pairs + e;
  const v2u64* sp = reinterpret_cast<const v2u64*>(shat) + e;
  Acc a0, a1;
  acc_zero(a0);
  acc_zero(a1);
  if (active) {
    u32 j = g;
    const size_t stride = (size_t)c * pairs;       // one step of this thread through j
    if constexpr (DBUF) {
      if (j + (U - 1) * c < k) {
        v2u64 x[U], y[U], xn[U], yn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          x[u] = __builtin_nontemporal_load(cp + (size_t)j * pairs + u * stride);
          y[u] = sp[(size_t)j * pairs + u * stride];
        }
        for (; j + (2 * U - 1) * c < k; j += U * c) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            xn[u] = __builtin_nontemporal_load(cp + (size_t)(j + U * c) * pairs + u * stride);
            yn[u] = sp[(size_t)(j + U * c) * pairs + u * stride];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            acc_mac_dev(a0, x[u].x, y[u].x);
            acc_mac_dev(a1, x[u].y, y[u].y);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) { x[u] = xn[u]; y[u] = yn[u]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc_mac_dev(a0, x[u].x, y[u].x);
          acc_mac_dev(a1, x[u].y, y[u].y);
        }
        j += U * c;
      }
    } else {
      for (; j + (U - 1) * c < k; j += U * c) {
        v2u64 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          x[u] = __builtin_nontemporal_load(cp + (size_t)j * pairs + u * stride);
          y[u] = sp[(size_t)j * pairs + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc_mac_dev(a0, x[u].x, y[u].x);
          acc_mac_dev(a1, x[u].y, y[u].y);
        }
      }
    }
    for (; j < k; j += c) {
      v2u64 x0 = cp[(size_t)j * pairs];
      v2u64 y0 = sp[(size_t)j * pairs];
      acc_mac_dev(a0, x0.x, y0.x); acc_mac_dev(a1, x0.y, y0.y);
    }
  }
  const u32 limb = active ? (2 * e) / ell : 0;
  const Mod m = mods[limb];
  v2u64 part;
  part.x = acc_reduce(a0, m);
  part.y = acc_reduce(a1, m);
  if (active) dl[threadIdx.x] = part;
  __syncthreads();
  if (active && g == 0) {
    v2u64 s = part;
    for (u32 w = 1; w < c; ++w) {
      v2u64 t = dl[w * chunk + el];
      s.x = addmod(s.x, t.x, m.q);
      s.y = addmod(s.y, t.y, m.q);
    }
    const size_t o = (size_t)d * pairs + e;
    v2u64 c2 = reinterpret_cast<const v2u64*>(c2col)[o];
    s.x = submod(s.x, c2.x, m.q);
    s.y = submod(s.y, c2.y, m.q);
    reinterpret_cast<v2u64*>(noisy)[o] = s;
  }
}

// decrypt_mac, dealer-grouped form: one workgroup serves DG dealers, so every s-hat pair fetched
// (through L2) is used DG times and the vector-memory instruction count per streamed byte drops
// from 2 to 1 + 1/DG.  Thread (g, e) as above; UJ j-steps are issued together.
// gridDim.y > 1 (both forms): the k terms are cut into gridDim.y ranges; a workgroup then leaves the partial sum of
// its range in partial[range][dealer] (no c2) and decrypt_finish_kernel adds the ranges up -- small batches
// (a single decrypt_party_value is ONE workgroup otherwise) then spread over the chip.
template <int DG, int UJ, int MAXT>
__global__ __launch_bounds__(MAXT) void decrypt_mac_grouped_kernel(const u64* __restrict__ c1s,
                                                                    const u64* __restrict__ shat,
                                                                    const u64* __restrict__ c2col,
                                                                    u64* __restrict__ noisy,
                                                                    const Mod* __restrict__ mods, u32 k_all,
                                                                    u32 ell, u32 pairs, u32 c, u32 dealers, u64* __restrict__ partial) {
  extern __shared__ v2u64 dl[];
  const u32 kq = (k_all + gridDim.y - 1) / gridDim.y, jlo = blockIdx.y * kq;
  const u32 k = (jlo + kq) < k_all ? (jlo + kq) : k_all;       // this workgroup's terms: [jlo, k)
  const u32 d0 = blockIdx.x * DG;
  const u32 g = threadIdx.x / pairs, e = threadIdx.x % pairs;
  const bool active = threadIdx.x < c * pairs;
  const v2u64* sp = reinterpret_cast<const v2u64*>(shat) + e;
  const v2u64* cp[DG];
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) {
    const u32 d = (d0 + dd) < dealers ? (d0 + dd) : (dealers - 1);   // clamp: tail group re-reads the last dealer
    cp[dd] = reinterpret_cast<const v2u64*>(c1s) + (size_t)d * k_all * pairs + e;
  }
  Acc a0[DG], a1[DG];
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) { acc_zero(a0[dd]); acc_zero(a1[dd]); }
  if (active) {
    u32 j = jlo + g;
    for (; j + (UJ - 1) * c < k; j += UJ * c) {
      v2u64 y[UJ], x[UJ][DG];
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        y[u] = sp[(size_t)(j + u * c) * pairs];
#pragma unroll
        for (int dd = 0; dd < DG; ++dd) x[u][dd] = __builtin_nontemporal_load(cp[dd] + (size_t)(j + u * c) * pairs);
      }
#pragma unroll
      for (int u = 0; u < UJ; ++u)
#pragma unroll
        for (int dd = 0; dd < DG; ++dd) {
          acc_mac_dev(a0[dd], x[u][dd].x, y[u].x);
          acc_mac_dev(a1[dd], x[u][dd].y, y[u].y);
        }
    }
    for (; j < k; j += c) {
      v2u64 y0 = sp[(size_t)j * pairs];
#pragma unroll
      for (int dd = 0; dd < DG; ++dd) {
        v2u64 x0 = cp[dd][(size_t)j * pairs];
        acc_mac_dev(a0[dd], x0.x, y0.x);
        acc_mac_dev(a1[dd], x0.y, y0.y);
      }
    }
  }
  const u32 limb = active ? (2 * e) / ell : 0;
  const Mod m = mods[limb];
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) {
    v2u64 part;
    part.x = acc_reduce(a0[dd], m);
    part.y = acc_reduce(a1[dd], m);
    __syncthreads();
    if (active) dl[threadIdx.x] = part;
    __syncthreads();
    if (active && g == 0 && d0 + dd < dealers) {
      v2u64 sres = part;
      for (u32 w = 1; w < c; ++w) {
        v2u64 t = dl[w * pairs + e];
        sres.x = addmod(sres.x, t.x, m.q);
        sres.y = addmod(sres.y, t.y, m.q);
      }
      const size_t o = (size_t)(d0 + dd) * pairs + e;
      if (gridDim.y > 1) {
        reinterpret_cast<v2u64*>(partial)[(size_t)blockIdx.y * dealers * pairs + o] = sres;
      } else {
        v2u64 c2 = reinterpret_cast<const v2u64*>(c2col)[o];
        sres.x = submod(sres.x, c2.x, m.q);
        sres.y = submod(sres.y, c2.y, m.q);
        reinterpret_cast<v2u64*>(noisy)[o] = sres;
      }
    }
  }
}

// decrypt_mac, full-width form.  A wave-wide load moves at most 1 KiB and the chip sustains a fixed
// number of them per second, so a polynomial of L*l/2 = 64*FW + rem slot pairs is split into FW waves
// that own 64 pairs each (every load full width) plus ONE remainder wave whose lanes cover
// G = 64/rem' consecutive j at once (rem' = rem rounded up to a power of two): its loads are G
// segments of rem' pairs, again (nearly) full width.  The grouped form above leaves 15 % (L*l/2 = 272)
// to 47 % (68) of the lanes of its last wave idle on every load.  Two dealers per workgroup.
// (88 VGPRs is a budget, not an accident: 13 of these waves and the 8 waves of a decode workgroup share a CU when
// pvw_decrypt_batch_device overlaps the two; prefetching c2 at the top costs 12 registers, gains 1.5 % alone and
// loses 20 % overlapped.)
template <int DG, int UJ, int SB = 0>
__global__ __launch_bounds__(1024) void decrypt_mac_fw_kernel(const u64* __restrict__ c1s,
                                                               const u64* __restrict__ shat,
                                                               const u64* __restrict__ c2col,
                                                               u64* __restrict__ noisy,
                                                               const Mod* __restrict__ mods, u32 k_all, u32 ell,
                                                               u32 pairs, u32 FW, u32 cfull, u32 remp, u32 crem,
                                                               u32 dealers, u64* __restrict__ partial) {
  extern __shared__ v2u64 dl[];                        // [DG][waves*64] partial sums
  const u32 kq = (k_all + gridDim.y - 1) / gridDim.y, jlo = blockIdx.y * kq;
  const u32 k = (jlo + kq) < k_all ? (jlo + kq) : k_all;       // this workgroup's terms: [jlo, k)
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 nfull = cfull * FW, rem = pairs - FW * 64;
  const bool is_full = wave < nfull;                   // wave-uniform
  const u32 d0 = blockIdx.x * DG;
  u32 e, jstart, jstep;
  bool active;
  if (is_full) {
    const u32 g = wave / FW, fw = wave % FW;
    e = fw * 64 + lane;
    jstart = g;
    jstep = cfull;
    active = true;
  } else {
    const u32 rw = wave - nfull;                       // replica among the remainder waves
    const u32 G = 64 / remp, jsub = lane / remp, er = lane % remp;
    e = FW * 64 + er;
    jstart = rw * G + jsub;
    jstep = crem * G;
    active = er < rem;
  }
  const v2u64* sp = reinterpret_cast<const v2u64*>(shat) + (active ? e : 0);
  const v2u64* cp[DG];
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) {
    const u32 d = (d0 + dd) < dealers ? (d0 + dd) : (dealers - 1);
    cp[dd] = reinterpret_cast<const v2u64*>(c1s) + (size_t)d * k_all * pairs + (active ? e : 0);
  }
  Acc a0[DG], a1[DG];
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) { acc_zero(a0[dd]); acc_zero(a1[dd]); }
  if (active) {
    u32 j = jlo + jstart;
    for (; j + (UJ - 1) * jstep < k; j += UJ * jstep) {
      v2u64 y[UJ], x[UJ][DG];
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        y[u] = sp[(size_t)(j + u * jstep) * pairs];
#pragma unroll
        for (int dd = 0; dd < DG; ++dd) x[u][dd] = __builtin_nontemporal_load(cp[dd] + (size_t)(j + u * jstep) * pairs);
      }
      if (SB) __builtin_amdgcn_sched_barrier(0);                   // every load of the group issued before its first MAC
#pragma unroll
      for (int u = 0; u < UJ; ++u)
#pragma unroll
        for (int dd = 0; dd < DG; ++dd) {
          acc_mac_dev(a0[dd], x[u][dd].x, y[u].x);
          acc_mac_dev(a1[dd], x[u][dd].y, y[u].y);
        }
      if (SB) __builtin_amdgcn_sched_barrier(0);
    }
    for (; j < k; j += jstep) {
      v2u64 y0 = sp[(size_t)j * pairs];
#pragma unroll
      for (int dd = 0; dd < DG; ++dd) {
        v2u64 x0 = cp[dd][(size_t)j * pairs];
        acc_mac_dev(a0[dd], x0.x, y0.x);
        acc_mac_dev(a1[dd], x0.y, y0.y);
      }
    }
  }
  const u32 limb = active ? (2 * e) / ell : 0;
  const Mod m = mods[limb];
  const u32 T = blockDim.x;
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) {
    v2u64 part;
    part.x = acc_reduce(a0[dd], m);
    part.y = acc_reduce(a1[dd], m);
    dl[dd * T + threadIdx.x] = part;
  }
  __syncthreads();
  // one owner per pair sums the partials of its replicas and finishes: full waves of replica 0 and
  // the lanes with jsub == 0 of remainder replica 0
  bool owner;
  if (is_full) owner = wave < FW;
  else owner = active && wave == nfull && lane < remp;
  if (owner) {
#pragma unroll
    for (int dd = 0; dd < DG; ++dd) {
      if (d0 + dd >= dealers) continue;
      v2u64 sres = (v2u64){0, 0};
      if (is_full) {
        for (u32 g = 0; g < cfull; ++g) {
          v2u64 tq = dl[dd * T + (g * FW + wave) * 64 + lane];
          sres.x = addmod(sres.x, tq.x, m.q);
          sres.y = addmod(sres.y, tq.y, m.q);
        }
      } else {
        const u32 G = 64 / remp;
        for (u32 rw = 0; rw < crem; ++rw)
          for (u32 js = 0; js < G; ++js) {
            v2u64 tq = dl[dd * T + (nfull + rw) * 64 + js * remp + lane];
            sres.x = addmod(sres.x, tq.x, m.q);
            sres.y = addmod(sres.y, tq.y, m.q);
          }
      }
      const size_t o = (size_t)(d0 + dd) * pairs + e;
      if (gridDim.y > 1) {
        reinterpret_cast<v2u64*>(partial)[(size_t)blockIdx.y * dealers * pairs + o] = sres;
      } else {
        v2u64 c2 = reinterpret_cast<const v2u64*>(c2col)[o];
        sres.x = submod(sres.x, c2.x, m.q);
        sres.y = submod(sres.y, c2.y, m.q);
        reinterpret_cast<v2u64*>(noisy)[o] = sres;
      }
    }
  }
}

// decrypt_finish: noisy[d] = INTT( sum_r partial[r][d] - c2col[d] )   (decryption.rs:268-274 and the
// change_representation(PowerBasis) of :116): the range sums of a split decrypt_mac are added up where the inverse
// transform reads them anyway.  ELL/2 threads per (dealer, limb): thread b adds up slots 2b, 2b+1 of the ranges,
// subtracts c2 and takes one butterfly per stage of ntt_inverse through LDS (the ELL/2 threads are consecutive and share
// a wave, whose LDS accesses execute in order: no barrier).  A split decrypt is a SMALL batch -- one decrypt_party_value
// is 34 such polynomials -- and with one thread per polynomial this pass was 17.6 us of pure latency at the config-5
// geometry.
template <int ELL>
__global__ __launch_bounds__(256) void decrypt_finish_kernel(const u64* __restrict__ partial, u32 nsplit,
                                                              const u64* __restrict__ c2col, u64* __restrict__ noisy,
                                                              u32 dealers, u32 L, DevTables t) {
  constexpr u32 H = ELL / 2;
  __shared__ u64 buf[256 / H * ELL];
  const u32 tid = blockIdx.x * 256 + threadIdx.x;
  const u32 pl = tid / H, b = tid % H;                   // (dealer, limb) pair; butterfly
  const bool on = pl < dealers * L;
  const u32 limb = on ? pl % L : 0;
  const Mod m = t.mods[limb];
  const size_t o = (size_t)(on ? pl : 0) * ELL + 2 * b, plane = (size_t)dealers * L * ELL;
  u64* a = buf + (threadIdx.x / H) * ELL;
  if (on) {
    const v2u64 c2 = *reinterpret_cast<const v2u64*>(c2col + o);
    v2u64 acc = *reinterpret_cast<const v2u64*>(partial + o);
    for (u32 r = 1; r < nsplit; ++r) {
      const v2u64 p = *reinterpret_cast<const v2u64*>(partial + r * plane + o);
      acc.x = addmod(acc.x, p.x, m.q);
      acc.y = addmod(acc.y, p.y, m.q);
    }
    a[2 * b] = submod(acc.x, c2.x, m.q);
    a[2 * b + 1] = submod(acc.y, c2.y, m.q);
  }
  __builtin_amdgcn_wave_barrier();
  const u64* tw = t.itw + (size_t)limb * ELL;
  const u64* twp = t.itwp + (size_t)limb * ELL;
  u32 step = 1;
#pragma unroll
  for (u32 mm = H; mm >= 1; mm >>= 1) {
    const u32 i = b / step, j = 2 * i * step + (b % step);
    if (on) {
      const u64 u = a[j], v = a[j + step];
      a[j] = addmod(u, v, m.q);
      a[j + step] = mulmod_shoup(submod(u, v, m.q), tw[mm + i], twp[mm + i], m.q);
    }
    __builtin_amdgcn_wave_barrier();
    step <<= 1;
  }
  if (on) {
    const u64 li = t.linv[limb], lip = t.linvp[limb];
    *reinterpret_cast<v2u64*>(noisy + o) = (v2u64){mulmod_shoup(a[2 * b], li, lip, m.q), mulmod_shoup(a[2 * b + 1], li, lip, m.q)};
  }
}
// how many ranges of j a decrypt over `dealers` ciphertexts is cut into (1 = no split): enough workgroups to put one
// on every CU when the batch alone does not, never ranges shorter than 64 terms
u32 decrypt_split(u32 k, u32 L, u32 ell, size_t dealers) {
  const u32 pairs = L * ell / 2;
  if (pairs > 1024 || dealers == 0) return 1;            // the generic form does not split
  u32 ns = (u32)PVW_ENV_INT("PVW_DEC_SPLIT", 0);          // tuning build: forced split
  if (ns == 0) {
    // measured at config 5 (profiles/r02_decrypt_split.txt): once every CU has a workgroup, cutting the ranges only
    // costs (356 -> 370 / 377 / 387 us at 2 / 4 / 8 ranges); the split is for small batches -- a single
    // decrypt_party_value is one workgroup streaming k polynomials alone otherwise
    const size_t wgs = (dealers + 1) / 2;
    ns = wgs >= 256 ? 1 : (u32)((256 + wgs - 1) / wgs);
    if (ns > 8) ns = 8;
  }
  while (ns > 1 && (k + ns - 1) / ns < 64) --ns;
  return ns ? ns : 1;
}

hipError_t launch_decrypt_finish(const u64* partial, u32 nsplit, const u64* c2col, u64* noisy, const DevTables& t, u32 L, u32 ell,
                                 size_t dealers, hipStream_t s) {
  if (dealers == 0) return hipSuccess;
  const size_t threads = dealers * L * (ell / 2);
  PVW_DISPATCH_ELL(ell, decrypt_finish_kernel<E><<<dim3((u32)((threads + 255) / 256)), dim3(256), 0, s>>>(partial, nsplit, c2col, noisy, (u32)dealers, L, t));
  return hipGetLastError();
}

hipError_t launch_decrypt_mac(const u64* c1s, const u64* shat, const u64* c2col, u64* noisy,
                              const DevTables& t, u32 k, u32 L, u32 ell, size_t dealers,
                              hipStream_t s, u64* partial, u32 nsplit, bool alone) {
  if (dealers == 0) return hipSuccess;
  if (nsplit == 0 || !partial) nsplit = 1;
  const u32 pairs = L * ell / 2;
  u32 step, c, threads, ny;
  if (pairs <= 1024) {
    step = pairs;
    ny = 1;
    c = pairs >= 256 ? 1 : 256 / pairs;
    if (c > k) c = k;
    threads = ((c * pairs + 63) / 64) * 64;
  } else {
    step = 1024;
    ny = (pairs + 1023) / 1024;
    c = 1;
    threads = 1024;
  }
  // by shape (profiles/r01_variant_sweep.txt, profiles/r01d_decrypt_sweep.txt): the full-width form from 128 slot pairs per
  // polynomial up, the dealer-grouped form below that, one workgroup per (dealer, 1024 pairs) beyond 1024 pairs.
  // Tuning build: PVW_DEC_VARIANT 10 = grouped / 60 = full-width where the shape allows, PVW_DEC_C = j-replicas
  // (read on every launch so that the tests can walk them in one process).
  int variant = (int)PVW_ENV_INT("PVW_DEC_VARIANT", 0), cenv = (int)PVW_ENV_INT("PVW_DEC_C", 0);
  if (variant == 0) variant = pairs >= 128 && pairs <= 1024 ? 60 : 10;
  if (variant < 60 && pairs <= 1024 && cenv > 0 && (u32)cenv * pairs <= 1024 && (u32)cenv <= k) {
    c = (u32)cenv;
    threads = ((c * pairs + 63) / 64) * 64;
  }
  const size_t lds = (size_t)threads * sizeof(v2u64);
  if (variant >= 60 && pairs <= 1024) {
    const u32 FW = pairs / 64, rem = pairs % 64;
    u32 remp = 0;
    if (rem) { remp = 1; while (remp < rem) remp <<= 1; }
    // replicas of the full waves (each takes every cfull-th j): the largest ODD count that fits 16 waves.
    // Measured at l=16, L=34 (272 pairs): 1, 2 replicas 398-402 us, 3 replicas 357 us; even counts lose on
    // every shape tried, and one 13-wave workgroup per CU beats two 5-wave ones.
    const u32 has_rem = rem ? 1u : 0u;
    u32 cfull = 0;
    if (FW) {
      cfull = (16 - has_rem) / FW;
      if (cfull > 1 && cfull % 2 == 0) --cfull;
      if (cfull > 7) cfull = 7;
      if (cenv > 0 && (u32)cenv * FW + has_rem <= 16) cfull = (u32)cenv;
      if (cfull > k) cfull = k;
    }
    const u32 crem = rem ? (FW ? 1 : 4) : 0;
    const u32 waves = cfull * FW + crem;
    if (waves >= 1 && waves <= 16) {
      const u32 thr = waves * 64;
      // two dealers per workgroup, four j-steps in flight (one / three dealers and 2 / 8 steps were measured: r01d_decrypt_sweep;
      // again in round 3 with the register budget varied as well, profiles/r03_decrypt_mac_shapes.txt: 425-735 us against 362-369)
      // Left to itself the compiler sinks each j-step's three loads down to their MACs (3 KiB in flight per wave, 88 registers).
      // SB = 1 keeps the group's twelve loads together in front of its MACs (124 registers): 362 vs 370 us at the config-5
      // shard.  Not when a decode is to share the CUs with this launch (the overlapped batch path): 13 of these waves and the
      // 8 of a decode workgroup fit a CU only at 88 registers.
      if (alone)
        decrypt_mac_fw_kernel<2, 4, 1><<<dim3((u32)((dealers + 1) / 2), nsplit), dim3(thr), (size_t)2 * thr * sizeof(v2u64), s>>>(
            c1s, shat, c2col, noisy, t.mods, k, ell, pairs, FW, cfull, remp, crem, (u32)dealers, partial);
      else
        decrypt_mac_fw_kernel<2, 4><<<dim3((u32)((dealers + 1) / 2), nsplit), dim3(thr), (size_t)2 * thr * sizeof(v2u64), s>>>(
            c1s, shat, c2col, noisy, t.mods, k, ell, pairs, FW, cfull, remp, crem, (u32)dealers, partial);
      return hipGetLastError();
    }
  }
  if (variant >= 10 && ny == 1) {
    // dealer-grouped form: two dealers per workgroup, two j-steps in flight
    if (threads <= 512)
      decrypt_mac_grouped_kernel<2, 2, 512><<<dim3((u32)((dealers + 1) / 2), nsplit), dim3(threads), lds, s>>>(
          c1s, shat, c2col, noisy, t.mods, k, ell, pairs, c, (u32)dealers, partial);
    else
      decrypt_mac_grouped_kernel<2, 2, 1024><<<dim3((u32)((dealers + 1) / 2), nsplit), dim3(threads), lds, s>>>(
          c1s, shat, c2col, noisy, t.mods, k, ell, pairs, c, (u32)dealers, partial);
    return hipGetLastError();
  }
  for (size_t off = 0; off < dealers; off += 32768) {
    const u32 nd = (u32)((dealers - off) < 32768 ? (dealers - off) : 32768);
    const u64* c1p = c1s + off * (size_t)k * L * ell;
    const u64* c2p = c2col + off * (size_t)L * ell;
    u64* np = noisy + off * (size_t)L * ell;
    decrypt_mac_kernel<4, true><<<dim3(nd, ny), dim3(threads), lds, s>>>(c1p, shat, c2p, np, t.mods, k, ell, pairs, c, step);
  }
  return hipGetLastError();
}

}  // namespace pvw
