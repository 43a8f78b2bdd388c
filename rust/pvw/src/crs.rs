//! `PvwCrs` constructors over `libpvw_hip.so` (NOT COMPILED here -- see rust/README.md).
//! Replaces the bodies of `src/params/crs.rs:24-90`; the struct keeps its fields (`matrix`, `params`), the matrix
//! additionally becomes resident on the device of `params.hip`.
use std::ffi::CString;
use std::sync::Arc;

use fhe_math::rq::Poly;
use ndarray::Array2;
use pvw_hip_sys as sys;
use rand::{CryptoRng, RngCore, SeedableRng};
use rand_chacha::ChaCha8Rng;

use super::parameters::{PvwParameters, Result};
use crate::errors::PvwError;
use crate::ffi_support::{check, poly_from_flat, poly_words, seed_from_rng};
use crate::params::crs::PvwCrs;

impl PvwCrs {
    /// `PvwCrs::new` (crs.rs:24-39): k x k uniform polynomials.  The RNG supplies the 32-byte seed of the device
    /// generator; nothing else about it crosses the boundary.
    pub fn new<R: RngCore + CryptoRng>(params: &Arc<PvwParameters>, rng: &mut R) -> Result<Self> {
        Self::new_deterministic(params, seed_from_rng(rng))
    }

    /// `PvwCrs::new_deterministic` (crs.rs:45-67): same seed, same CRS on every party.  The polynomials are this
    /// library's ChaCha8 streams (`PVW_DOM_CRS`), not `Poly::random_from_seed`'s bytes: parties must all use the
    /// same implementation, exactly as they must all use the same seed.
    pub fn new_deterministic(params: &Arc<PvwParameters>, seed: <ChaCha8Rng as SeedableRng>::Seed) -> Result<Self> {
        let ctx = params.hip.raw();
        check(unsafe { sys::pvw_crs_generate(ctx, seed.as_ptr()) })?;
        // host copy for `get`, `iter`, serde, ...: downloaded once, power basis, then NTT on the host side
        let k = params.k;
        let words = poly_words(params);
        let mut flat = vec![0u64; k * k * words];
        check(unsafe { sys::pvw_get_crs(ctx, flat.as_mut_ptr(), sys::PVW_REPR_POWER) })?;
        let polys: Result<Vec<Poly>> = flat.chunks_exact(words).map(|c| poly_from_flat(c, params)).collect();
        let matrix = Array2::from_shape_vec((k, k), polys?)
            .map_err(|_| PvwError::CrsError("CRS matrix has the wrong shape".to_string()))?;
        Ok(Self { matrix, params: params.clone() })
    }

    /// `PvwCrs::new_from_tag` (crs.rs:74-90): the seed is `DefaultHasher(tag + "CRS")`, eight little-endian bytes
    /// repeated four times -- computed by the library so that every host language derives the same bytes.
    pub fn new_from_tag(params: &Arc<PvwParameters>, tag: &str) -> Result<Self> {
        let c_tag = CString::new(tag).map_err(|_| PvwError::InvalidParameters("tag contains a NUL byte".to_string()))?;
        let mut seed = [0u8; 32];
        check(unsafe { sys::pvw_crs_seed_from_tag(c_tag.as_ptr(), seed.as_mut_ptr()) })?;
        Self::new_deterministic(params, seed)
    }

    /// A CRS obtained elsewhere (deserialised, or generated with fhe-math): make it resident.
    pub fn upload(&self) -> Result<()> {
        let mut flat = Vec::with_capacity(self.params.k * self.params.k * poly_words(&self.params));
        for poly in self.matrix.iter() {
            crate::ffi_support::poly_to_flat(poly, &mut flat);
        }
        check(unsafe { sys::pvw_load_crs(self.params.hip.raw(), flat.as_ptr(), sys::PVW_REPR_POWER) })
    }
}
