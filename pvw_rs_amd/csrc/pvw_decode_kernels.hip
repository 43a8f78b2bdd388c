// pvw_decode_kernels.hip -- decode_scalar_pvw_rns (decryption.rs:10-247) on gfx950: the gadget decode as fixed-width
// big-integer arithmetic, one 64-bit word per lane (decode_chain_kernel); a thread-per-ciphertext form (decode_kernel,
// pvw_decode.h) for modulus chains wider than a wave holds.
#include <hip/hip_runtime.h>

#include "pvw_arith.h"
#include "pvw_chacha.h"
#include "pvw_decode.h"
#include "pvw_kernels.h"
#include "pvw_dev.h"

#include "pvw_decode_wave.h"

namespace pvw {

// ------------------------------------------------------------------------------------
// decode: decode_scalar_pvw_rns (decryption.rs:10-58) on the device, one thread per ciphertext.
// Big integers live in LDS with the thread index as the fast axis (word j of thread t at [j][t]).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void decode_kernel(const u64* __restrict__ noisy, u64* __restrict__ out,
                                                     u32 count, DecodeTables t) {
  extern __shared__ u64 dsm[];
  const u32 d = blockIdx.x * 64 + threadIdx.x;
  u64* base = dsm + threadIdx.x;
  BN x{base, 64};
  BN y{base + (size_t)(t.W + 1) * 64, 64};
  BN nres{base + (size_t)(2 * t.W + 1) * 64, 64};
  if (d >= count) return;
  out[d] = decode_one_fixed(t, noisy + (size_t)d * t.L * t.ell, x, y, nres);
}

// decode, lifted-chain form (pvw_decode_wave.h): 4 waves per ciphertext, cpw ciphertexts per workgroup
// XF: the residues arrive in the NTT domain and are transformed back while they are staged (stage_inverse).  A separate
// instance because the compiler gives it 94 registers where the plain one has 79: at most 80 keep the decode co-resident
// with decrypt_mac_fw, which the overlapped batch path relies on -- that path uses the plain instance behind launch_ntt.
// wipe / wipe16: a region (16-byte units) the launch clears on its way -- NTT(sk) of the decrypt this decode closes, which every
// kernel in front of it on the stream has finished reading (the reference's SecretKey is ZeroizeOnDrop; a memset launch of
// its own cost 4 us + two launch gaps per call).  The grid may hold workgroups beyond `ndec` that do nothing else.
template <bool XF>
__global__ __launch_bounds__(512) void decode_chain_kernel(u64* __restrict__ noisy, u64* __restrict__ out,
                                                            u32 count, u32 cpw_dbg, DecodeTables t, InverseTables xf,
                                                            u64* __restrict__ wipe, u32 wipe16, u32 ndec) {
  extern __shared__ u64 dws[];
  if (wipe16) {
    v2u64* wp = reinterpret_cast<v2u64*>(wipe);
    for (u32 x = blockIdx.x * 512 + threadIdx.x; x < wipe16; x += gridDim.x * 512) wp[x] = (v2u64){0, 0};
  }
  if (blockIdx.x >= ndec) return;
  if (!XF) xf.itw = nullptr;
  decode_chain_body<4>(noisy, out, count, cpw_dbg, t, xf, blockIdx.x, dws);
}

// Kernels that may ask for more than the default 64 KiB of dynamic LDS.  The attribute is per device and per
// code object, so it is set once per CONTEXT while the context initialises its device (ensure_device, under the
// context's init mutex, after hipSetDevice) -- not lazily behind process-wide flags.
hipError_t init_kernel_attributes() {
  const int big = 160 * 1024;
  hipError_t e = hipFuncSetAttribute((const void*)decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)decode_chain_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)decode_chain_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
#if PVW_TUNING
  if (e == hipSuccess) e = init_probe_attributes();
#endif
  return e;
}

// LDS bytes of decode_chain_kernel for cpw ciphertexts of wpc waves per workgroup
static size_t decode_chain_lds(const DecodeTables& t, u32 cpw, u32 wpc) {
  return ((size_t)t.L * t.W + 2 * (2 * t.W + 2) + 256 + (size_t)cpw * ((size_t)(t.ell + 1) * 64 + (size_t)t.L * t.ell + (size_t)5 * t.ell + 2)) * 8;
}

hipError_t launch_decode(u64* noisy, u64* out, size_t count, const DecodeTables& t, hipStream_t s, const DevTables* xf,
                         u64* wipe, size_t wipe_bytes, bool* wiped) {
  if (wiped) *wiped = false;
  if (count == 0) return hipSuccess;
  InverseTables inv{nullptr, nullptr, nullptr, nullptr};
  if (xf) inv = InverseTables{xf->itw, xf->itwp, xf->linv, xf->linvp};
  // by shape: the lifted chain (4 waves per ciphertext, 2 ciphertexts per workgroup) while L <= 64 and W + 2 <= 63
  // (Q up to ~3900 bits) and its tables fit the LDS; one thread per ciphertext beyond (tuning build: PVW_DECODE_VARIANT=1
  // forces it).
  const int variant = (int)PVW_ENV_INT("PVW_DECODE_VARIANT", 0);
  if (variant != 1 && t.L <= 64 && t.W + 2 <= 63) {
    const u32 wpc = 4, cpw = 2;
    const size_t bytes = decode_chain_lds(t, cpw, wpc);
    if (bytes <= 160 * 1024) {
      const u32 ndec = (u32)((count + cpw - 1) / cpw);
      const bool do_wipe = wipe && wipe_bytes && wipe_bytes % 16 == 0 && wipe_bytes / 16 < (1ull << 32);
      const u32 wipe16 = do_wipe ? (u32)(wipe_bytes / 16) : 0;
      const dim3 grid(do_wipe && ndec < 128 ? 128u : ndec), block(cpw * wpc * 64);      // a small batch gets helpers for the wipe
      if (wiped) *wiped = do_wipe;
      // tuning build only: PVW_DECODE_TIMING=1..6: out[] = cycles of a phase (results are NOT values; tools/decode_timing.py)
      const u32 dbg = (u32)PVW_ENV_INT("PVW_DECODE_TIMING", 0);
      const u32 no_small = PVW_ENV_INT("PVW_DECODE_SMALL", 1) == 0 ? 1u << 31 : 0;     // tuning build: every lift in full
      if (xf) decode_chain_kernel<true><<<grid, block, bytes, s>>>(noisy, out, (u32)count, cpw | ((dbg & 0xff) << 16) | no_small, t, inv, wipe, wipe16, ndec);
      else decode_chain_kernel<false><<<grid, block, bytes, s>>>(noisy, out, (u32)count, cpw | ((dbg & 0xff) << 16) | no_small, t, inv, wipe, wipe16, ndec);
      return hipGetLastError();
    }
  }
  if (xf) {                                              // the fixed-width form has no transform of its own
    hipError_t e = launch_ntt(noisy, count, true, *xf, t.L, t.ell, s);
    if (e != hipSuccess) return e;
  }
  const size_t lds = (size_t)(2 * t.W + 1 + t.L) * 64 * sizeof(u64);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  decode_kernel<<<dim3((u32)((count + 63) / 64)), dim3(64), lds, s>>>(noisy, out, (u32)count, t);
  return hipGetLastError();
}

}  // namespace pvw
