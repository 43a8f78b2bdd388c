/* pvw_hip_tuning.h -- entry points that exist ONLY in the measurement build libpvw_hip_tuning.so
 * (hipcc -DPVW_TUNING=1, pvw_rs_amd/build.py).  That build also honours the environment switches listed in
 * DESIGN.md section 7a (kernel schedule selectors such as PVW_MAC_VARIANT / PVW_MAC_PACKED / PVW_DEC_VARIANT,
 * and the timing aids PVW_DECODE_TIMING / PVW_GEMM_ZERO_OPERANDS, which produce WRONG results by design).
 * The shipped libpvw_hip.so exports none of this and reads no environment variable: the reference samples and
 * computes unconditionally (src/crypto/encryption.rs:135-167), and so must its drop-in.
 * the tools/ scripts, bench.py's read_probe leg and tests/test_gpu_tuning.py load the tuning build; nothing else does. */
#ifndef PVW_HIP_TUNING_H
#define PVW_HIP_TUNING_H

#include "pvw_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* MEASUREMENT AID (bench.py): seconds per pass of a read-only kernel with the access pattern of the streamed
 * inner products (crs.rs:188-201, encryption.rs:177-200) over the resident public-key section */
PVW_API int32_t pvw_selftest_read_bandwidth(pvw_ctx* ctx, uint32_t reps, double* seconds_per_pass, uint64_t* bytes_per_pass);

/* MEASUREMENT AID (tools/probe_sweep.py): the read probe with `u` tiles of 1 KiB (2u when dbuf & 1) in flight per wave
 * and lds_bytes of unused LDS per workgroup, which caps the workgroups resident per CU; dbuf & 2: every XCD walks a
 * contiguous eighth of the matrix instead of every eighth 256-KiB run */
PVW_API int32_t pvw_tuning_read_probe(pvw_ctx* ctx, uint32_t reps, uint32_t u, uint32_t dbuf, uint32_t lds_bytes,
                                      double* seconds_per_pass, uint64_t* bytes_per_pass);
/* MEASUREMENT AID (tools/mac_timeline.py): with PVW_MAC_VARIANT=40 (tiled stream) or 44 (packed stream) every workgroup of the
 * streamed-inner-product kernel records its first and last instruction on the constant 100 MHz counter; this reads
 * them back: stamps [count][2], hw_id [count] (HW_ID register of the workgroup's first wave: XCC, SE, CU). */
PVW_API int32_t pvw_tuning_read_stamps(pvw_ctx* ctx, uint64_t* stamps, uint32_t* hw_id, uint32_t count);

#ifdef __cplusplus
}
#endif
#endif /* PVW_HIP_TUNING_H */
