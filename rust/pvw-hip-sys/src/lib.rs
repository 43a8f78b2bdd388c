//! Raw bindings to `libpvw_hip.so` -- one to one with `include/pvw_hip.h`.
//!
//! **NOT COMPILED in this repository's pipeline** (no Rust toolchain in the image; see `rust/README.md`).
//! `tests/test_rust_binding.py` keeps this file honest: it parses the header and the `extern "C"` block below and
//! asserts the same symbol set, arity and argument widths, the `#[repr(C)]` field order and the status codes.
//!
//! Conventions (details in the header): every function returns `i32` (0 = `PVW_OK`, otherwise the 1-based index of
//! the `PvwError` variant in declaration order, `src/errors.rs:13-70`; the message is `pvw_last_error`, thread
//! local); a polynomial is `[L][l]` `u64`, limb-major (`src/params/parameters.rs:433-458`); host-buffer calls are
//! synchronous and may be issued concurrently on one context (rayon does, `src/crypto/encryption.rs:277-283`);
//! `*_device` calls take device pointers and a `hipStream_t` and only enqueue.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_void};

/// Opaque device context behind one `PvwParameters` (`pvw_ctx` in the header).
#[repr(C)]
pub struct PvwCtx {
    _private: [u8; 0],
}

/// Opaque device-resident secret key (`pvw_sk` in the header): NTT(sk) as the inner products of decrypt read it.
#[repr(C)]
pub struct PvwSk {
    _private: [u8; 0],
}

/// `pvw_params_t`: the `PvwParametersBuilder` fields (`src/params/parameters.rs:44-52`) plus device placement.
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct PvwParamsT {
    pub n: u32,
    pub k: u32,
    pub l: u32,
    pub num_moduli: u32,
    pub moduli: *const u64,
    pub secret_variance: f32,
    pub error_bound_1: u64,
    pub error_bound_2: u64,
    pub device: i32,
    pub party_lo: u32,
    pub party_hi: u32,
    pub c1_lo: u32,
    pub c1_hi: u32,
}

/// `pvw_randomness_t`: the randomness of one encrypt call (the reference draws from `thread_rng()` inside rayon
/// closures, `src/crypto/encryption.rs:138,164,180`, which cannot cross an FFI boundary).
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct PvwRandomnessT {
    pub mode: u32,
    pub seed: [u8; 32],
    pub r: *const i64,
    pub e1: *const i64,
    pub e2: *const i64,
}

pub const PVW_PREPARE_PACKED: u32 = 1;
pub const PVW_PREPARE_MFMA: u32 = 2;
pub const PVW_OK: i32 = 0;
pub const PVW_ERR_INVALID_PARAMETERS: i32 = 1;
pub const PVW_ERR_SAMPLING: i32 = 2;
pub const PVW_ERR_ENCRYPTION: i32 = 3;
pub const PVW_ERR_DECRYPTION: i32 = 4;
pub const PVW_ERR_KEY_GENERATION: i32 = 5;
pub const PVW_ERR_CRS: i32 = 6;
pub const PVW_ERR_SERIALIZATION: i32 = 7;
pub const PVW_ERR_DESERIALIZATION: i32 = 8;
pub const PVW_ERR_ENCODING: i32 = 9;
pub const PVW_ERR_DECODING: i32 = 10;
pub const PVW_ERR_VALIDATION: i32 = 11;
pub const PVW_ERR_CONTEXT: i32 = 12;
pub const PVW_ERR_POLYNOMIAL: i32 = 13;
pub const PVW_ERR_MATRIX: i32 = 14;
pub const PVW_ERR_DIMENSION_MISMATCH: i32 = 15;
pub const PVW_ERR_INDEX_OUT_OF_BOUNDS: i32 = 16;
pub const PVW_ERR_INSUFFICIENT_DATA: i32 = 17;
pub const PVW_ERR_INVALID_FORMAT: i32 = 18;
pub const PVW_ERR_INTERNAL: i32 = 19;

pub const PVW_REPR_POWER: u32 = 0;
pub const PVW_REPR_NTT: u32 = 1;
pub const PVW_RND_SEED: u32 = 0;
pub const PVW_RND_EXPLICIT: u32 = 1;

pub const PVW_DOM_R: u32 = 0;
pub const PVW_DOM_E1: u32 = 1;
pub const PVW_DOM_E2: u32 = 2;
pub const PVW_DOM_SK: u32 = 3;
pub const PVW_DOM_EKEY: u32 = 4;
pub const PVW_DOM_CRS: u32 = 5;
pub const PVW_DOM_GAUSS: u32 = 6;
pub const PVW_DOM_PK: u32 = 7;

extern "C" {
    // ---- errors / device ------------------------------------------------------------------
    pub fn pvw_last_error(buf: *mut c_char, len: usize) -> i32;
    pub fn pvw_device_available() -> i32;
    // ---- parameters: PvwParametersBuilder::build (src/params/parameters.rs:117-195) --------
    pub fn pvw_ctx_create(params: *const PvwParamsT, out: *mut *mut PvwCtx) -> i32;
    pub fn pvw_ctx_destroy(ctx: *mut PvwCtx) -> i32;
    pub fn pvw_ctx_get_roots(ctx: *const PvwCtx, psi_out: *mut u64) -> i32;
    pub fn pvw_ctx_set_roots(ctx: *mut PvwCtx, psi: *const u64) -> i32;
    pub fn pvw_ctx_delta(ctx: *const PvwCtx, words: *mut u64, cap: usize, nwords: *mut usize) -> i32;
    pub fn pvw_ctx_delta_power_l_minus_1(ctx: *const PvwCtx, words: *mut u64, cap: usize, nwords: *mut usize) -> i32;
    pub fn pvw_ctx_q_total(ctx: *const PvwCtx, words: *mut u64, cap: usize, nwords: *mut usize) -> i32;
    pub fn pvw_ctx_gadget(ctx: *const PvwCtx, poly_out: *mut u64, repr: u32) -> i32;
    pub fn pvw_ctx_verify_correctness_condition(ctx: *const PvwCtx, ok_out: *mut i32) -> i32;
    pub fn pvw_suggest_error_bounds(n: u32, k: u32, l: u32, moduli: *const u64, num_moduli: u32, variance: f32, bound1_out: *mut u32, bound2_out: *mut u32) -> i32;
    pub fn pvw_encode_scalar(ctx: *const PvwCtx, scalar: i64, poly_out: *mut u64, repr: u32) -> i32;
    // ---- CRS: PvwCrs.matrix (src/params/crs.rs:12-17), constructors :24-90 -------------------
    pub fn pvw_load_crs(ctx: *mut PvwCtx, a: *const u64, repr: u32) -> i32;
    pub fn pvw_load_crs_device(ctx: *mut PvwCtx, d_a: *const u64, repr: u32, stream: *mut c_void) -> i32;
    pub fn pvw_crs_generate(ctx: *mut PvwCtx, seed: *const u8) -> i32;
    pub fn pvw_crs_seed_from_tag(tag: *const c_char, seed_out: *mut u8) -> i32;
    pub fn pvw_get_crs(ctx: *mut PvwCtx, a_out: *mut u64, repr: u32) -> i32;
    // ---- global public key: GlobalPublicKey.matrix (src/keys/public_key.rs:43-54, :214-250) --
    pub fn pvw_load_pk(ctx: *mut PvwCtx, party_lo: u32, party_hi: u32, b: *const u64, repr: u32) -> i32;
    pub fn pvw_load_pk_device(ctx: *mut PvwCtx, party_lo: u32, party_hi: u32, d_b: *const u64, repr: u32, stream: *mut c_void) -> i32;
    pub fn pvw_pk_fill_uniform(ctx: *mut PvwCtx, seed: *const u8) -> i32;
    pub fn pvw_get_pk(ctx: *mut PvwCtx, party_lo: u32, party_hi: u32, b_out: *mut u64, repr: u32) -> i32;
    pub fn pvw_num_public_keys(ctx: *const PvwCtx, out: *mut u32) -> i32;
    pub fn pvw_is_full(ctx: *const PvwCtx, out: *mut i32) -> i32;
    // ---- key generation: PublicKey::generate (public_key.rs:111-147) over crs.rs:138-171 ------
    pub fn pvw_keygen(ctx: *mut PvwCtx, party_lo: u32, party_hi: u32, sk: *const i64, ek: *const i64, seed: *const u8) -> i32;
    pub fn pvw_sample_secret_keys(ctx: *const PvwCtx, seed: *const u8, party_lo: u32, count: u32, sk_out: *mut i64) -> i32;
    // ---- encrypt (src/crypto/encryption.rs:105-214), encrypt_all_party_shares (:253-286) ------
    pub fn pvw_encrypt(ctx: *mut PvwCtx, scalars: *const u64, num_scalars: usize, rnd: *const PvwRandomnessT, c1_out: *mut u64, c2_out: *mut u64, out_repr: u32) -> i32;
    pub fn pvw_encrypt_device(ctx: *mut PvwCtx, d_scalars: *const u64, num_scalars: usize, rnd: *const PvwRandomnessT, d_c1: *mut u64, d_c2: *mut u64, out_repr: u32, stream: *mut c_void) -> i32;
    pub fn pvw_encrypt_multi(ctx: *mut PvwCtx, scalars: *const u64, num_dealers: usize, scalars_per_dealer: usize, seeds: *const u8, c1_out: *mut u64, c2_out: *mut u64, out_repr: u32) -> i32;
    pub fn pvw_encrypt_multi_device(ctx: *mut PvwCtx, d_scalars: *const u64, num_dealers: usize, scalars_per_dealer: usize, seeds: *const u8, d_c1: *mut u64, d_c2: *mut u64, out_repr: u32, stream: *mut c_void) -> i32;
    // ---- decrypt (src/crypto/decryption.rs:249-325) and gadget decode (:10-247) ---------------
    pub fn pvw_decrypt_batch(ctx: *mut PvwCtx, sk: *const i64, c1s: *const u64, c2col: *const u64, num_dealers: usize, in_repr: u32, out_u64: *mut u64, noisy_out: *mut u64) -> i32;
    pub fn pvw_decrypt_noisy_device(ctx: *mut PvwCtx, d_sk: *const i64, d_c1s: *const u64, d_c2col: *const u64, num_dealers: usize, in_repr: u32, d_noisy: *mut u64, stream: *mut c_void) -> i32;
    pub fn pvw_decrypt_batch_device(ctx: *mut PvwCtx, d_sk: *const i64, d_c1s: *const u64, d_c2col: *const u64, num_dealers: usize, in_repr: u32, d_noisy: *mut u64, d_out: *mut u64, stream: *mut c_void) -> i32;
    pub fn pvw_sk_load(ctx: *mut PvwCtx, sk: *const i64, out: *mut *mut PvwSk) -> i32;
    pub fn pvw_sk_free(key: *mut PvwSk) -> i32;
    pub fn pvw_decrypt_batch_device_sk(ctx: *mut PvwCtx, key: *const PvwSk, d_c1s: *const u64, d_c2col: *const u64, num_dealers: usize, in_repr: u32, d_noisy: *mut u64, d_out: *mut u64, stream: *mut c_void) -> i32;
    pub fn pvw_decode(ctx: *mut PvwCtx, noisy: *const u64, count: usize, out_u64: *mut u64) -> i32;
    pub fn pvw_decode_host(ctx: *const PvwCtx, noisy: *const u64, count: usize, out_u64: *mut u64) -> i32;
    pub fn pvw_decode_device(ctx: *mut PvwCtx, d_noisy: *const u64, count: usize, d_out: *mut u64, stream: *mut c_void) -> i32;
    // ---- self-tests / build identity -------------------------------------------------------------
    pub fn pvw_selftest_decode_fixed(ctx: *const PvwCtx, noisy: *const u64, count: usize, out_u64: *mut u64) -> i32;
    pub fn pvw_selftest_mfma_i8(ctx: *mut PvwCtx, a: *const i8, b: *const i8, out: *mut i32) -> i32;
    pub fn pvw_selftest_secret_residue(ctx: *mut PvwCtx, nonzero_words: *mut u64, scanned_words: *mut u64) -> i32;
    pub fn pvw_selftest_siphash(msg: *const u8, len: usize, k0: u64, k1: u64, c_rounds: i32, d_rounds: i32, out: *mut u64) -> i32;
    pub fn pvw_selftest_decode_tables(ctx: *const PvwCtx, info_out: *mut u32) -> i32;
    pub fn pvw_selftest_decode_shortcuts(ctx: *const PvwCtx, noisy: *const u64, count: usize, out_u64: *mut u64, short_path: *mut u8) -> i32;
    pub fn pvw_build_is_tuning() -> i32;
    // ---- ring primitives (fhe-math call sites: change_representation, from_coefficients) ---------
    pub fn pvw_ntt_forward(ctx: *mut PvwCtx, polys: *mut u64, count: usize) -> i32;
    pub fn pvw_ntt_inverse(ctx: *mut PvwCtx, polys: *mut u64, count: usize) -> i32;
    pub fn pvw_small_to_poly(ctx: *mut PvwCtx, coeffs: *const i64, count: usize, polys: *mut u64, repr: u32) -> i32;
    // ---- samplers (src/sampling/uniform.rs:5-70, normal.rs:12-20,136-190) ------------------------
    pub fn pvw_sample_cbd(ctx: *mut PvwCtx, seed: *const u8, domain: u32, index0: u32, count: usize, variance: f32, out: *mut i64) -> i32;
    pub fn pvw_sample_uniform(ctx: *mut PvwCtx, seed: *const u8, domain: u32, index0: u32, count: usize, bound: u64, out: *mut i64) -> i32;
    pub fn pvw_sample_gaussian(ctx: *mut PvwCtx, seed: *const u8, index0: u32, count: usize, bound: u64, out: *mut i64) -> i32;
    // ---- measurement hooks -----------------------------------------------------------------------
    pub fn pvw_ctx_set_profiling(ctx: *mut PvwCtx, on: i32) -> i32;
    pub fn pvw_ctx_kernel_time(ctx: *mut PvwCtx, name: *const c_char, total_ms: *mut f64, launches: *mut u64) -> i32;
    pub fn pvw_ctx_reset_profiling(ctx: *mut PvwCtx) -> i32;
    pub fn pvw_host_alloc(bytes: usize, out: *mut *mut c_void) -> i32;
    pub fn pvw_host_free(p: *mut c_void) -> i32;
    pub fn pvw_ctx_resident_bytes(ctx: *const PvwCtx, crs_bytes: *mut u64, pk_bytes: *mut u64) -> i32;
    pub fn pvw_ctx_derived_bytes(ctx: *const PvwCtx, packed_bytes: *mut u64, mfma_tiled_bytes: *mut u64) -> i32;
    pub fn pvw_prepare(ctx: *mut PvwCtx, flags: u32, stream: *mut c_void, bytes_out: *mut u64) -> i32;
    pub fn pvw_ctx_packed_active(ctx: *const PvwCtx, width_out: *mut u32) -> i32;
    pub fn pvw_ctx_synchronize(ctx: *mut PvwCtx) -> i32;
}
