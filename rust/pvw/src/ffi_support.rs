//! Glue between the reference's types and `libpvw_hip.so` (NOT COMPILED here -- see rust/README.md).
//!
//! * `check`     : status code -> `PvwError`, all 19 variants (`src/errors.rs:13-70`)
//! * `HipContext`: one `pvw_ctx` per `PvwParameters` (created in `PvwParametersBuilder::build`,
//!                 `src/params/parameters.rs:117-195`; `Arc<PvwParameters>` already shares it)
//! * flat `[L][l]` u64 <-> `fhe_math::rq::Poly`, through the constructor the reference uses itself
//!   (`Poly::try_convert_from(Array2<u64>, ctx, false, PowerBasis)`, `src/params/parameters.rs:453-466`)
use std::ffi::CStr;
use std::sync::Arc;

use fhe_math::rq::{Poly, Representation};
use ndarray::Array2;
use pvw_hip_sys as sys;

use crate::errors::{PvwError, PvwResult};
use crate::params::PvwParameters;

/// The two integers a `{expected, actual}` / `{index, bound}` message carries ("expected 4, got 3",
/// "Index 7 out of bounds for 4 polynomials", "7 >= 4"): the first two runs of digits.
fn two_numbers(msg: &str) -> (usize, usize) {
    let mut it = msg
        .split(|c: char| !c.is_ascii_digit())
        .filter(|s| !s.is_empty())
        .filter_map(|s| s.parse::<usize>().ok());
    (it.next().unwrap_or(0), it.next().unwrap_or(0))
}

/// Status code of a C-ABI call -> the reference's error type.  Code i (1..=19) is the i-th variant of `PvwError` in
/// declaration order; the message is the calling thread's `pvw_last_error`.
pub fn check(rc: i32) -> PvwResult<()> {
    if rc == sys::PVW_OK {
        return Ok(());
    }
    let mut buf = [0 as std::os::raw::c_char; 512];
    unsafe { sys::pvw_last_error(buf.as_mut_ptr(), buf.len()) };
    let msg = unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();
    Err(match rc {
        sys::PVW_ERR_INVALID_PARAMETERS => PvwError::InvalidParameters(msg),
        sys::PVW_ERR_SAMPLING => PvwError::SamplingError(msg),
        sys::PVW_ERR_ENCRYPTION => PvwError::EncryptionError(msg),
        sys::PVW_ERR_DECRYPTION => PvwError::DecryptionError(msg),
        sys::PVW_ERR_KEY_GENERATION => PvwError::KeyGenerationError(msg),
        sys::PVW_ERR_CRS => PvwError::CrsError(msg),
        sys::PVW_ERR_SERIALIZATION => PvwError::SerializationError(msg),
        sys::PVW_ERR_DESERIALIZATION => PvwError::DeserializationError(msg),
        sys::PVW_ERR_ENCODING => PvwError::EncodingError(msg),
        sys::PVW_ERR_DECODING => PvwError::DecodingError(msg),
        sys::PVW_ERR_VALIDATION => PvwError::ValidationError(msg),
        sys::PVW_ERR_CONTEXT => PvwError::ContextError(msg),
        sys::PVW_ERR_POLYNOMIAL => PvwError::PolynomialError(msg),
        sys::PVW_ERR_MATRIX => PvwError::MatrixError(msg),
        sys::PVW_ERR_DIMENSION_MISMATCH => {
            let (expected, actual) = two_numbers(&msg);
            PvwError::DimensionMismatch { expected, actual }
        }
        sys::PVW_ERR_INDEX_OUT_OF_BOUNDS => {
            let (index, bound) = two_numbers(&msg);
            PvwError::IndexOutOfBounds { index, bound }
        }
        sys::PVW_ERR_INSUFFICIENT_DATA => {
            let (expected, actual) = two_numbers(&msg);
            PvwError::InsufficientData { expected, actual }
        }
        sys::PVW_ERR_INVALID_FORMAT => PvwError::InvalidFormat(msg),
        _ => PvwError::InternalError(msg), // 19, and anything a newer library might add
    })
}

/// Owner of the device context of one parameter set.  A new field `hip: HipContext` of `PvwParameters`
/// (`src/params/parameters.rs:19-40`), filled in by `build()`.
#[derive(Debug)]
pub struct HipContext {
    ctx: *mut sys::PvwCtx,
}
// the library serialises what needs it internally; host-buffer calls are safe from many threads
unsafe impl Send for HipContext {}
unsafe impl Sync for HipContext {}

impl HipContext {
    /// Called at the end of `PvwParametersBuilder::build` with the validated fields.  `u64` bounds: the reference
    /// holds BigInt bounds (`parameters.rs:91-104`); values that do not fit are rejected here.
    pub fn new(n: usize, k: usize, l: usize, moduli: &[u64], secret_variance: f32, bound1: u64, bound2: u64) -> PvwResult<Self> {
        let p = sys::PvwParamsT {
            n: n as u32,
            k: k as u32,
            l: l as u32,
            num_moduli: moduli.len() as u32,
            moduli: moduli.as_ptr(),
            secret_variance,
            error_bound_1: bound1,
            error_bound_2: bound2,
            device: -1,
            party_lo: 0,
            party_hi: 0,
            c1_lo: 0,
            c1_hi: 0,
        };
        let mut ctx: *mut sys::PvwCtx = std::ptr::null_mut();
        check(unsafe { sys::pvw_ctx_create(&p, &mut ctx) })?;
        Ok(Self { ctx })
    }
    pub fn raw(&self) -> *mut sys::PvwCtx {
        self.ctx
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        unsafe { sys::pvw_ctx_destroy(self.ctx) };
    }
}

/// Words of one polynomial: `num_moduli * l`.
pub fn poly_words(params: &PvwParameters) -> usize {
    params.context.moduli().len() * params.l
}

/// `[L][l]` power-basis residues -> `Poly` in NTT representation (the representation every `Poly` of the reference's
/// hot path is kept in).  The two libraries' NTT-domain layouts are not interchangeable, so data crosses in the
/// power basis (include/pvw_hip.h, `repr`).
pub fn poly_from_flat(flat: &[u64], params: &Arc<PvwParameters>) -> PvwResult<Poly> {
    let rows = params.context.moduli().len();
    let m = Array2::from_shape_vec((rows, params.l), flat.to_vec())
        .map_err(|_| PvwError::PolynomialError("flat polynomial has the wrong length".to_string()))?;
    let mut poly = Poly::try_convert_from(m, &params.context, false, Representation::PowerBasis)
        .map_err(|e| PvwError::PolynomialError(format!("Failed to create polynomial from RNS coefficients: {e:?}")))?;
    poly.change_representation(Representation::Ntt);
    Ok(poly)
}

/// `Poly` (any representation) -> `[L][l]` power-basis residues appended to `out`.
pub fn poly_to_flat(poly: &Poly, out: &mut Vec<u64>) {
    let mut p = poly.clone();
    if *p.representation() != Representation::PowerBasis {
        p.change_representation(Representation::PowerBasis);
    }
    out.extend(p.coefficients().iter().copied()); // ArrayView2<u64>, shape (num_moduli, degree), row-major
}

/// A 32-byte seed from the caller's RNG: what replaces handing the RNG itself across the boundary.
pub fn seed_from_rng<R: rand::RngCore + rand::CryptoRng>(rng: &mut R) -> [u8; 32] {
    let mut seed = [0u8; 32];
    rng.fill_bytes(&mut seed);
    seed
}
