// pvw_poly.hip -- the small-polynomial kernels of the PVW path on gfx950: signed coefficients -> RNS -> l-point NTT
// (prep, prologue), in-place (I)NTT, API layout <-> tiled matrix, the samplers.  All O(n + k) polynomials per call:
// launch-latency sized.  Where the transform comes with other per-polynomial work (sampling, tiling) it is one thread per
// (polynomial, limb), fully unrolled in registers; the plain transforms (ntt_kernel, prep_coop_kernel) take l/2 threads
// per polynomial, one butterfly per thread and stage through LDS.
#include <hip/hip_runtime.h>

#include "pvw_arith.h"
#include "pvw_chacha.h"
#include "pvw_decode.h"
#include "pvw_kernels.h"
#include "pvw_dev.h"

namespace pvw {

// ------------------------------------------------------------------------------------
// prep: small signed coefficients -> RNS -> l-point NTT (+ scalar * g-hat), one thread per
// (polynomial, limb).  Serves r-hat, the e1/e2 addends, encode_scalar
// (src/params/parameters.rs:346-367) and Poly::from_coefficients + NTT
// (encryption.rs:147-154, secret_key.rs:98-112).
// ------------------------------------------------------------------------------------
template <int ELL>
__global__ __launch_bounds__(64) void prep_kernel(const i64* __restrict__ coeffs,
                                                   const u64* __restrict__ scalars,
                                                   u64* __restrict__ out, size_t stride_poly,
                                                   size_t stride_limb, u32 count, u32 L,
                                                   u32 do_ntt, DevTables t, u32 group, size_t stride_group) {
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= count * L) return;
  const u32 p = tid / L, limb = tid % L;
  const Mod m = t.mods[limb];
  u64 a[ELL];
#pragma unroll
  for (int s = 0; s < ELL; ++s) a[s] = signed_residue(coeffs[(size_t)p * ELL + s], m);
  if (do_ntt) ntt_forward<ELL>(a, t.tw + (size_t)limb * ELL, t.twp + (size_t)limb * ELL, m);
  if (scalars) {
    // `scalars[i] as i64` wrap (encryption.rs:195), then scalar * g  (parameters.rs:346-367)
    const u64 mr = signed_residue((i64)scalars[p], m);
    const u64* g = (do_ntt ? t.ghat : t.gpow) + (size_t)limb * ELL;
    const u64* gp = (do_ntt ? t.ghatp : t.gpowp) + (size_t)limb * ELL;
#pragma unroll
    for (int s = 0; s < ELL; ++s) a[s] = addmod(a[s], mulmod_shoup(mr, g[s], gp[s], m.q), m.q);
  }
  u64* o = out + (group ? (size_t)(p / group) * stride_group + (size_t)(p % group) * stride_poly : (size_t)p * stride_poly) +
           (size_t)limb * stride_limb;
#pragma unroll
  for (int s = 0; s < ELL; s += 2)
    *reinterpret_cast<v2u64*>(o + s) = (v2u64){a[s], a[s + 1]};
}

// The same for the plain case (small coefficients -> RNS -> NTT, no scalar, no grouping) with ELL/2 threads per
// (polynomial, limb) instead of one: thread b brings in coefficients 2b, 2b+1, takes one butterfly per stage of
// ntt_forward through LDS and stores slots 2b, 2b+1.  The ELL/2 threads are consecutive and share a wave, whose LDS
// accesses execute in order: no barrier between the stages.  The transform of a decrypt's secret key is k x L short
// transforms in front of the inner products: with one thread each the launch was 10.5 us of latency at config 5.
template <int ELL>
__global__ __launch_bounds__(256) void prep_coop_kernel(const i64* __restrict__ coeffs, u64* __restrict__ out,
                                                         size_t stride_poly, size_t stride_limb, u32 count, u32 L, DevTables t) {
  constexpr u32 H = ELL / 2;
  __shared__ u64 buf[256 / H * ELL];
  const u32 tid = blockIdx.x * 256 + threadIdx.x;
  const u32 pl = tid / H, b = tid % H;                   // (polynomial, limb) pair; butterfly
  const bool on = pl < count * L;
  const u32 p = on ? pl / L : 0, limb = on ? pl % L : 0;
  const Mod m = t.mods[limb];
  u64* a = buf + (threadIdx.x / H) * ELL;
  if (on) {
    a[2 * b] = signed_residue(coeffs[(size_t)p * ELL + 2 * b], m);
    a[2 * b + 1] = signed_residue(coeffs[(size_t)p * ELL + 2 * b + 1], m);
  }
  __builtin_amdgcn_wave_barrier();
  const u64* tw = t.tw + (size_t)limb * ELL;
  const u64* twp = t.twp + (size_t)limb * ELL;
  u32 step = ELL;
#pragma unroll
  for (u32 mm = 1; mm < ELL; mm <<= 1) {
    step >>= 1;
    const u32 i = b / step, j = 2 * i * step + (b % step);
    if (on) {
      const u64 u = a[j], v = mulmod_shoup(a[j + step], tw[mm + i], twp[mm + i], m.q);
      a[j] = addmod(u, v, m.q);
      a[j + step] = submod(u, v, m.q);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (on)
    *reinterpret_cast<v2u64*>(out + (size_t)p * stride_poly + (size_t)limb * stride_limb + 2 * b) = (v2u64){a[2 * b], a[2 * b + 1]};
}

// dst[c][j] = src[j][c] over a k x k matrix of polynomials (`words` u64 each): key generation walks the
// CRS by columns (crs.rs:152-168)
__global__ __launch_bounds__(256) void transpose_polys_kernel(const u64* __restrict__ src, u64* __restrict__ dst,
                                                               u32 k, u32 words) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)k * k * words) return;
  const u32 x = idx % words;
  const size_t pj = idx / words;
  const u32 j = pj / k, c = pj % k;
  dst[((size_t)c * k + j) * words + x] = src[idx];
}

// in-place change_representation on [count][L][l] polynomials, one thread per polynomial, fully unrolled in registers:
// the form for large batches (fewest instructions per polynomial)
template <int ELL>
__global__ __launch_bounds__(64) void ntt_poly_kernel(u64* __restrict__ polys, u32 count, u32 L,
                                                       u32 inverse, DevTables t) {
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= count * L) return;
  const u32 limb = tid % L;
  const Mod m = t.mods[limb];
  u64* p = polys + (size_t)tid * ELL;
  u64 a[ELL];
#pragma unroll
  for (int s = 0; s < ELL; s += 2) {
    v2u64 v = *reinterpret_cast<const v2u64*>(p + s);
    a[s] = v.x;
    a[s + 1] = v.y;
  }
  if (inverse) ntt_inverse<ELL>(a, t.itw + (size_t)limb * ELL, t.itwp + (size_t)limb * ELL, t.linv[limb], t.linvp[limb], m);
  else ntt_forward<ELL>(a, t.tw + (size_t)limb * ELL, t.twp + (size_t)limb * ELL, m);
#pragma unroll
  for (int s = 0; s < ELL; s += 2)
    *reinterpret_cast<v2u64*>(p + s) = (v2u64){a[s], a[s + 1]};
}

// the same for small batches (a decrypt's D polynomials, an encrypt's n + k): ELL/2 threads per polynomial, thread b brings in and
// takes out slots 2b, 2b+1 (16 contiguous bytes: a wave moves whole lines) and takes one butterfly per stage of
// ntt_forward / ntt_inverse through LDS; the ELL/2 threads are consecutive and share a wave, whose LDS accesses execute in
// order, so the stages need no barrier.  A batch this small is pure latency with one thread per polynomial (decrypt_finish:
// 17.6 -> 7 us); a large one is better off with fewer instructions per polynomial (config 5 in full, 8192 x 34
// polynomials per step beside the inner products: 145 us of transform time against 183 us in this form).
template <int ELL>
__global__ __launch_bounds__(256) void ntt_kernel(u64* __restrict__ polys, u32 count, u32 L,
                                                   u32 inverse, DevTables t) {
  constexpr u32 H = ELL / 2;
  __shared__ u64 buf[256 / H * ELL];
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t pl = tid / H;
  const u32 b = (u32)(tid % H);
  const bool on = pl < (size_t)count * L;
  const u32 limb = on ? (u32)(pl % L) : 0;
  const Mod m = t.mods[limb];
  u64* p = polys + (on ? pl : 0) * ELL + 2 * b;
  u64* a = buf + (threadIdx.x / H) * ELL;
  if (on) {
    const v2u64 v = *reinterpret_cast<const v2u64*>(p);
    a[2 * b] = v.x;
    a[2 * b + 1] = v.y;
  }
  __builtin_amdgcn_wave_barrier();
  if (inverse) {
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
      *reinterpret_cast<v2u64*>(p) = (v2u64){mulmod_shoup(a[2 * b], li, lip, m.q), mulmod_shoup(a[2 * b + 1], li, lip, m.q)};
    }
  } else {
    const u64* tw = t.tw + (size_t)limb * ELL;
    const u64* twp = t.twp + (size_t)limb * ELL;
    u32 step = ELL;
#pragma unroll
    for (u32 mm = 1; mm < ELL; mm <<= 1) {
      step >>= 1;
      const u32 i = b / step, j = 2 * i * step + (b % step);
      if (on) {
        const u64 u = a[j], v = mulmod_shoup(a[j + step], tw[mm + i], twp[mm + i], m.q);
        a[j] = addmod(u, v, m.q);
        a[j + step] = submod(u, v, m.q);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (on) *reinterpret_cast<v2u64*>(p) = (v2u64){a[2 * b], a[2 * b + 1]};
  }
}

// ------------------------------------------------------------------------------------
// tile / untile: API layout [row][j][L][l] <-> tiled M, optional NTT on the way.
// One thread per (row, j, limb).
// ------------------------------------------------------------------------------------
template <int ELL>
__global__ __launch_bounds__(256) void tile_kernel(const u64* __restrict__ src, u64* __restrict__ M,
                                                    u32 rows, u32 row0_tiled, u32 k, u32 L,
                                                    u32 ntt_first, DevTables t) {
  constexpr int R = 128 / ELL;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (size_t)rows * k * L) return;
  const u32 limb = tid % L;
  const u32 j = (tid / L) % k;
  const u32 row = tid / ((size_t)L * k);
  const u64* p = src + tid * ELL;
  u64 a[ELL];
#pragma unroll
  for (int s = 0; s < ELL; s += 2) {
    v2u64 v = *reinterpret_cast<const v2u64*>(p + s);
    a[s] = v.x;
    a[s + 1] = v.y;
  }
  if (ntt_first) ntt_forward<ELL>(a, t.tw + (size_t)limb * ELL, t.twp + (size_t)limb * ELL, t.mods[limb]);
  const u32 trow = row0_tiled + row;
  u64* o = M + (((size_t)(trow / R) * L + limb) * k + j) * 128 + (trow % R) * ELL;
#pragma unroll
  for (int s = 0; s < ELL; s += 2)
    *reinterpret_cast<v2u64*>(o + s) = (v2u64){a[s], a[s + 1]};
}

template <int ELL>
__global__ __launch_bounds__(256) void untile_kernel(const u64* __restrict__ M, u64* __restrict__ dst,
                                                      u32 rows, u32 row0_tiled, u32 k, u32 L,
                                                      u32 intt_after, DevTables t) {
  constexpr int R = 128 / ELL;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (size_t)rows * k * L) return;
  const u32 limb = tid % L;
  const u32 j = (tid / L) % k;
  const u32 row = tid / ((size_t)L * k);
  const u32 trow = row0_tiled + row;
  const u64* p = M + (((size_t)(trow / R) * L + limb) * k + j) * 128 + (trow % R) * ELL;
  u64 a[ELL];
#pragma unroll
  for (int s = 0; s < ELL; s += 2) {
    v2u64 v = *reinterpret_cast<const v2u64*>(p + s);
    a[s] = v.x;
    a[s + 1] = v.y;
  }
  if (intt_after) ntt_inverse<ELL>(a, t.itw + (size_t)limb * ELL, t.itwp + (size_t)limb * ELL, t.linv[limb], t.linvp[limb], t.mods[limb]);
  u64* o = dst + tid * ELL;
#pragma unroll
  for (int s = 0; s < ELL; s += 2)
    *reinterpret_cast<v2u64*>(o + s) = (v2u64){a[s], a[s + 1]};
}

// uniform residues straight into the tiled matrix: polynomial (grow, j), limb i uses ChaCha8
// stream (domain << 32) | ((grow*k + j)*L + i)   (grow = global row index)
template <int ELL>
__global__ __launch_bounds__(256) void fill_uniform_tiled_kernel(u64* __restrict__ M, ChaChaKey key,
                                                                  u32 domain, u32 rows,
                                                                  u32 row0_tiled, u32 grow0, u32 k,
                                                                  u32 L, DevTables t) {
  constexpr int R = 128 / ELL;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (size_t)rows * k * L) return;
  const u32 limb = tid % L;
  const u32 j = (tid / L) % k;
  const u32 row = tid / ((size_t)L * k);
  ChaChaRng g;
  g.init(key, domain, (u32)((((size_t)(grow0 + row)) * k + j) * L + limb));
  u64 a[ELL];
  const u64 q = t.mods[limb].q;
  const u32 sh = (u32)__clzll((long long)q);
#pragma unroll
  for (int s = 0; s < ELL; ++s) {
    u64 v;
    do { v = g.next_u64() >> sh; } while (v >= q);
    a[s] = v;
  }
  const u32 trow = row0_tiled + row;
  u64* o = M + (((size_t)(trow / R) * L + limb) * k + j) * 128 + (trow % R) * ELL;
#pragma unroll
  for (int s = 0; s < ELL; s += 2)
    *reinterpret_cast<v2u64*>(o + s) = (v2u64){a[s], a[s + 1]};
}

// ------------------------------------------------------------------------------------
// samplers: one thread per polynomial, coefficient order and word consumption as the
// reference's samplers (src/sampling/uniform.rs).  out [count][l] i64.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sample_kernel(i64* __restrict__ out, ChaChaKey key, u32 l,
                                                     SampleJob j0, SampleJob j1, SampleJob j2) {
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  SampleJob job;
  u32 local;
  if (tid < j0.count) { job = j0; local = tid; }
  else if (tid < j0.count + j1.count) { job = j1; local = tid - j0.count; }
  else if (tid < j0.count + j1.count + j2.count) { job = j2; local = tid - j0.count - j1.count; }
  else return;
  ChaChaRng g;
  g.init(key, job.domain, job.index0 + local);
  i64* o = out + ((size_t)job.out_poly0 + local) * l;
  auto emit = [o](u32 s, i64 v) { o[s] = v; };
  if (job.kind == SAMPLE_CBD) sample_cbd_poly(g, l, job.cbd_half != 0, job.cbd_v, emit);
  else sample_uniform_poly(g, l, job.bound, emit);
}

// ------------------------------------------------------------------------------------
// prologue: everything encrypt needs before the streamed MAC, in ONE launch
// (encryption.rs:135-154 r, :161-167 e1, :195-196 encode + e2): each block takes PB <= 64
// polynomials, samples (or copies) their small coefficients into LDS with one thread per
// polynomial, then one thread per (polynomial, limb) reduces, transforms and stores.
// ------------------------------------------------------------------------------------
// Latency is what this kernel is made of (one encrypt's worth is 4608 polynomials: a launch that cannot fill the chip
// for long), so the dependent memory round trips are counted: the batch descriptor travels in the kernel-argument
// segment (host memory behind PCIe unless the runtime keeps kernel arguments on the device -- every dependent read of
// it costs microseconds) and is therefore read ONCE, by one wide load per workgroup into LDS; job look-ups after
// that are LDS reads.  The inputs of the transform phase that live in device memory (the party's scalar, the limb's
// modulus) are requested before the sampling phase and arrive under it.
//   hop 1 scalar header (implicit)  ->  hop 2 descriptor -> LDS  ->  [ sampling || table staging, scalar / modulus
//   loads ]  ->  transform  ->  store
template <int ELL>
__global__ __launch_bounds__(256) void prologue_kernel(PrologueBatch b, u32 L, u32 PB, u32 stage_tables, DevTables t) {
  extern __shared__ u64 psm[];
  i64* sc = reinterpret_cast<i64*>(psm);              // [PB][ELL] sampled coefficients
  u64* tab = psm + (size_t)PB * ELL;                  // [4][L][ELL] tw | twp | ghat | ghatp (if staged)
  constexpr u32 JOB_WORDS = sizeof(PrologueJob) / 4, KEY_WORDS = sizeof(ChaChaKey) / 4;
  static_assert(sizeof(PrologueJob) % 8 == 0, "descriptor copy is word-wise");
  u32* jobw = reinterpret_cast<u32*>(tab + (stage_tables ? (size_t)4 * L * ELL : 0));   // [njobs] PrologueJob
  u32* keyw = jobw + PVW_MAX_PROLOGUE_JOBS * JOB_WORDS;                                  // [key_window] ChaChaKey
  const PrologueJob* jobs = reinterpret_cast<const PrologueJob*>(jobw);
  const ChaChaKey* keys = reinterpret_cast<const ChaChaKey*>(keyw);
  const u32 gp0 = blockIdx.x * PB;
  const u32 tid = threadIdx.x;
  const u32 rep = blockIdx.y;                         // replica (dealer / party) of the template jobs
  // ---- the descriptor: one coalesced read of the jobs and of this replica's key window ----
  {
    const u32* src = reinterpret_cast<const u32*>(&b.job[0]);
    const u32 nw = b.njobs * JOB_WORDS;
    for (u32 w = tid; w < nw; w += 256) jobw[w] = src[w];
    const u32* ksrc = reinterpret_cast<const u32*>(&b.key[rep * b.key_rep]);
    const u32 kw = b.key_window * KEY_WORDS;
    for (u32 w = tid; w < kw; w += 256) keyw[w] = ksrc[w];
  }
  if (stage_tables && tid >= 64) {
    // the three waves that do not sample bring the twiddle / gadget tables into LDS
    const u32 n = L * ELL;
    for (u32 x = tid - 64; x < n; x += 192) {
      tab[x] = t.tw[x];
      tab[n + x] = t.twp[x];
      tab[2 * n + x] = t.ghat[x];
      tab[3 * n + x] = t.ghatp[x];
    }
  }
  __syncthreads();
  // locate (job, local polynomial) of global polynomial gp: jobs are laid end to end
  auto locate = [&](u32 gp, u32& ji, u32& local) {
    ji = 0;
    local = gp;
#pragma unroll
    for (u32 x = 0; x + 1 < PVW_MAX_PROLOGUE_JOBS; ++x)
      if (ji == x && x + 1 < b.njobs && local >= jobs[x].sj.count) { local -= jobs[x].sj.count; ji = x + 1; }
  };
  // ---- this thread's (polynomial, limb) of the transform phase (first trip): request what it needs from device
  // memory now, so that it arrives while wave 0 samples ----
  const u32 p0 = tid / L, limb0 = tid % L;
  const bool work0 = tid < PB * L && gp0 + p0 < b.total;
  u32 ji0 = 0, local0 = 0;
  Mod m0 = Mod{1, 0, 0};
  u64 scalar0 = 0;
  if (work0) {
    locate(gp0 + p0, ji0, local0);
    m0 = t.mods[limb0];
    if (jobs[ji0].scalars && !jobs[ji0].raw_out) scalar0 = jobs[ji0].scalars[(size_t)rep * jobs[ji0].rep_scalars + local0];
  }
  if (tid < PB && tid < 64 && gp0 + tid < b.total) {
    u32 ji, local;
    locate(gp0 + tid, ji, local);
    const PrologueJob& job = jobs[ji];
    i64* o = job.raw_out ? job.raw_out + (size_t)local * ELL : sc + tid * ELL;    // sampled-only families go straight to memory
    if (job.explicit_coeffs) {
      const i64* ec = job.explicit_coeffs + (size_t)rep * job.rep_coeffs;
#pragma unroll
      for (int s = 0; s < ELL; ++s) o[s] = ec[(size_t)local * ELL + s];
    } else {
      ChaChaRng g;
      g.init(keys[job.key_idx], job.sj.domain, job.sj.index0 + rep * job.rep_index0 + local);
      auto emit = [o](u32 s, i64 v) { o[s] = v; };
      if (job.sj.kind == SAMPLE_CBD) sample_cbd_poly(g, ELL, job.sj.cbd_half != 0, job.sj.cbd_v, emit);
      else sample_uniform_poly(g, ELL, job.sj.bound, emit);
    }
  }
  __syncthreads();
  const u32 n = L * ELL;
  // one thread per (polynomial, limb); a block of PB <= 64 polynomials takes ceil(PB * L / 256) trips
  for (u32 idx = tid; idx < PB * L; idx += 256) {
    const u32 p = idx / L, limb = idx % L;
    if (gp0 + p >= b.total) break;
    u32 ji = ji0, local = local0;
    Mod m = m0;
    u64 scalar = scalar0;
    if (idx != tid) {                                    // later trips (PB * L > 256): the same look-ups, not prefetched
      locate(gp0 + p, ji, local);
      m = t.mods[limb];
      scalar = jobs[ji].scalars ? jobs[ji].scalars[(size_t)rep * jobs[ji].rep_scalars + local] : 0;
    }
    const PrologueJob& job = jobs[ji];
    if (job.raw_out) continue;                           // nothing to transform
    const u64* tw = stage_tables ? tab + (size_t)limb * ELL : t.tw + (size_t)limb * ELL;
    const u64* twp = stage_tables ? tab + n + (size_t)limb * ELL : t.twp + (size_t)limb * ELL;
    u64 a[ELL];
#pragma unroll
    for (int s = 0; s < ELL; ++s) a[s] = signed_residue(sc[p * ELL + s], m);
    ntt_forward<ELL>(a, tw, twp, m);
    if (job.scalars) {
      const u64 mr = signed_residue((i64)scalar, m);     // `as i64` wrap, encryption.rs:195
      const u64* g = stage_tables ? tab + 2 * n + (size_t)limb * ELL : t.ghat + (size_t)limb * ELL;
      const u64* gp = stage_tables ? tab + 3 * n + (size_t)limb * ELL : t.ghatp + (size_t)limb * ELL;
#pragma unroll
      for (int s = 0; s < ELL; ++s) a[s] = addmod(a[s], mulmod_shoup(mr, g[s], gp[s], m.q), m.q);
    }
    u64* o = job.out + (size_t)rep * job.rep_out + (size_t)local * job.stride_poly + (size_t)limb * job.stride_limb;
#pragma unroll
    for (int s = 0; s < ELL; s += 2) *reinterpret_cast<v2u64*>(o + s) = (v2u64){a[s], a[s + 1]};
  }
}

// truncated discrete Gaussian (src/sampling/normal.rs:136-190): one thread per sample.
__device__ __forceinline__ double unit_f64(ChaChaRng& g) {
  return (double)(g.next_u64() >> 11) * (1.0 / 9007199254740992.0);
}
__global__ __launch_bounds__(64) void gaussian_kernel(i64* __restrict__ out, ChaChaKey key,
                                                       u32 index0, u32 count, u64 bound) {
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= count) return;
  ChaChaRng g;
  g.init(key, DOM_GAUSS, index0 + tid);
  if (bound == 0) { out[tid] = 0; return; }            // normal.rs:137-139
  const double bf = (double)bound;
  if (bf > 1e15) {                                      // :144-149
    const i64 sign = (g.next_u32() >> 31) ? 1 : -1;
    out[tid] = sign * (i64)(g.next_u32() % 1000001u);
    return;
  }
  const double sigma = bf / 16.96;                      // :8,:151
  double ratio = 0.0;
  bool have = false;
  if (sigma > 0.3) {                                    // :168-170
    ratio = 2.0 * unit_f64(g) - 1.0;
    have = true;
  } else {
    for (int it = 0; it < 1000 && !have; ++it) {        // :173-179
      const double eps = 2.220446049250313e-16;
      const double u1 = eps + (1.0 - eps) * unit_f64(g);
      const double u2 = unit_f64(g);
      const double z = sqrt(-2.0 * log(u1)) * cos(2.0 * 3.14159265358979323846 * u2);  // :186-190
      const double r = z * sigma;
      if (r >= -1.0 && r <= 1.0) { ratio = r; have = true; }
    }
    if (!have) ratio = 2.0 * unit_f64(g) - 1.0;         // :182
  }
  const double fx = ratio * bf;                         // ratio_to_bigint fast path :199-204
  i64 x = (i64)floor(fabs(fx) + 0.5);
  if (fx < 0) x = -x;
  const i64 b = (i64)bound;
  out[tid] = x > b ? b : (x < -b ? -b : x);             // :156-160
}
hipError_t launch_prep(const i64* coeffs, const u64* scalars, u64* out, size_t stride_poly,
                       size_t stride_limb, u32 count, bool do_ntt, const DevTables& t, u32 L,
                       u32 ell, hipStream_t s, u32 group, size_t stride_group) {
  if (count == 0) return hipSuccess;
  if (do_ntt && !scalars && !group) {
    const size_t coop = (size_t)count * L * (ell / 2);
    PVW_DISPATCH_ELL(ell, prep_coop_kernel<E><<<dim3((u32)((coop + 255) / 256)), dim3(256), 0, s>>>(coeffs, out, stride_poly, stride_limb, count, L, t));
    return hipGetLastError();
  }
  const u32 threads = count * L;
  PVW_DISPATCH_ELL(ell, prep_kernel<E><<<dim3((threads + 63) / 64), dim3(64), 0, s>>>(coeffs, scalars, out, stride_poly, stride_limb, count, L,
                                            do_ntt ? 1u : 0u, t, group, stride_group));
  return hipGetLastError();
}

hipError_t launch_transpose_polys(const u64* src, u64* dst, u32 k, u32 words, hipStream_t s) {
  const size_t total = (size_t)k * k * words;
  if (total == 0) return hipSuccess;
  transpose_polys_kernel<<<dim3((u32)((total + 255) / 256)), dim3(256), 0, s>>>(src, dst, k, words);
  return hipGetLastError();
}

hipError_t launch_ntt(u64* polys, size_t count, bool inverse, const DevTables& t, u32 L, u32 ell,
                      hipStream_t s) {
  // keep each launch below 2^31 threads
  const size_t step = (size_t)1 << 24;
  for (size_t off = 0; off < count; off += step) {
    const u32 cnt = (u32)((count - off) < step ? (count - off) : step);
    u64* p = polys + off * L * ell;
    if ((size_t)cnt * L > 8192) {
      const u32 polys_l = cnt * L;
      PVW_DISPATCH_ELL(ell, ntt_poly_kernel<E><<<dim3((polys_l + 63) / 64), dim3(64), 0, s>>>(p, cnt, L, inverse ? 1u : 0u, t));
      continue;
    }
    const size_t threads = (size_t)cnt * L * (ell / 2);
    PVW_DISPATCH_ELL(ell, ntt_kernel<E><<<dim3((u32)((threads + 255) / 256)), dim3(256), 0, s>>>(p, cnt, L, inverse ? 1u : 0u, t));
  }
  return hipGetLastError();
}

hipError_t launch_tile(const u64* src, u64* M, u32 rows, u32 row0_tiled, u32 k, u32 L, u32 ell,
                       bool ntt_first, const DevTables& t, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  const size_t threads = (size_t)rows * k * L;
  PVW_DISPATCH_ELL(ell, tile_kernel<E><<<dim3((u32)((threads + 255) / 256)), dim3(256), 0, s>>>(src, M, rows, row0_tiled, k, L, ntt_first ? 1u : 0u, t));
  return hipGetLastError();
}

hipError_t launch_untile(const u64* M, u64* dst, u32 rows, u32 row0_tiled, u32 k, u32 L, u32 ell,
                         bool intt_after, const DevTables& t, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  const size_t threads = (size_t)rows * k * L;
  PVW_DISPATCH_ELL(ell, untile_kernel<E><<<dim3((u32)((threads + 255) / 256)), dim3(256), 0, s>>>(M, dst, rows, row0_tiled, k, L, intt_after ? 1u : 0u, t));
  return hipGetLastError();
}

hipError_t launch_fill_uniform_tiled(u64* M, const ChaChaKey& key, u32 domain, u32 rows,
                                     u32 row0_tiled, u32 grow0, u32 k, u32 L, u32 ell,
                                     const DevTables& t, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  const size_t threads = (size_t)rows * k * L;
  PVW_DISPATCH_ELL(ell, fill_uniform_tiled_kernel<E><<<dim3((u32)((threads + 255) / 256)), dim3(256), 0, s>>>(M, key, domain, rows, row0_tiled, grow0, k, L, t));
  return hipGetLastError();
}

hipError_t launch_sample(i64* out, const ChaChaKey& key, u32 ell, const SampleJob& j0,
                         const SampleJob& j1, const SampleJob& j2, hipStream_t s) {
  const u32 threads = j0.count + j1.count + j2.count;
  if (threads == 0) return hipSuccess;
  sample_kernel<<<dim3((threads + 63) / 64), dim3(64), 0, s>>>(out, key, ell, j0, j1, j2);
  return hipGetLastError();
}

hipError_t launch_prologue(const PrologueBatch& batch, const DevTables& t, u32 L, u32 ell, hipStream_t s) {
  PrologueBatch b = batch;
  b.total = 0;
  if (b.njobs > PVW_MAX_PROLOGUE_JOBS) return hipErrorInvalidValue;
  if (b.reps == 0) b.reps = 1;
  if (b.reps > 65535) return hipErrorInvalidValue;
  b.key_window = 1;
  b.key_rep = b.njobs ? b.job[0].rep_key : 0;
  for (u32 i = 0; i < b.njobs; ++i) {
    b.total += b.job[i].sj.count;
    if (b.job[i].key_idx + (b.reps - 1) * b.job[i].rep_key >= PVW_MAX_PROLOGUE_KEYS) return hipErrorInvalidValue;
    if (b.job[i].rep_key != b.key_rep) return hipErrorInvalidValue;     // one key policy per batch: shared, or one per replica
    if (b.job[i].key_idx + 1 > b.key_window) b.key_window = b.job[i].key_idx + 1;
  }
  if (b.key_window > 8) return hipErrorInvalidValue;
  if (b.total == 0) return hipSuccess;
  if (L > 256) return hipErrorInvalidValue;
  // polynomials per block: 256/L (one trip of the transform loop, lowest latency) for one encrypt's worth of
  // work; a whole wave of samplers (64) when the launch is large enough to fill the chip anyway
  u32 PB = 256 / L;
  if (PB > 64) PB = 64;
  if ((size_t)b.total * b.reps >= 65536) PB = 64;
  const u32 blocks = (b.total + PB - 1) / PB;
  const size_t sc_bytes = (size_t)PB * ell * 8, tab_bytes = (size_t)4 * L * ell * 8;
  const size_t desc_bytes = (size_t)PVW_MAX_PROLOGUE_JOBS * sizeof(PrologueJob) + 8 * sizeof(ChaChaKey);
  const u32 stage = (sc_bytes + tab_bytes + desc_bytes <= 64 * 1024) ? 1u : 0u;
  const size_t lds = sc_bytes + (stage ? tab_bytes : 0) + desc_bytes;
  PVW_DISPATCH_ELL(ell, prologue_kernel<E><<<dim3(blocks, b.reps), dim3(256), lds, s>>>(b, L, PB, stage, t));
  return hipGetLastError();
}

hipError_t launch_gaussian(i64* out, const ChaChaKey& key, u32 index0, u32 count, u64 bound,
                           hipStream_t s) {
  if (count == 0) return hipSuccess;
  gaussian_kernel<<<dim3((count + 63) / 64), dim3(64), 0, s>>>(out, key, index0, count, bound);
  return hipGetLastError();
}

}  // namespace pvw
static_assert(sizeof(pvw::PrologueBatch) <= 4000, "PrologueBatch must fit the kernel-argument segment");
