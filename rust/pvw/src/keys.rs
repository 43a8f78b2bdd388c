//! Key generation over `libpvw_hip.so` (NOT COMPILED here -- see rust/README.md).
//! Replaces the bodies of `PublicKey::generate` (`src/keys/public_key.rs:111-147`, over
//! `PvwCrs::multiply_by_secret_key`, `src/params/crs.rs:138-171`), `GlobalPublicKey::add_public_key` (:214-250),
//! `generate_and_add_party` (:256-263) and `generate_all_party_keys` (:376-401).
use fhe_math::rq::Poly;
use pvw_hip_sys as sys;
use rand::{CryptoRng, RngCore};
use zeroize::Zeroize;

use crate::errors::PvwError;
use crate::ffi_support::{check, poly_from_flat, poly_to_flat, poly_words, seed_from_rng};
use crate::keys::public_key::{GlobalPublicKey, Party, PublicKey};
use crate::keys::secret_key::SecretKey;
use crate::params::{PvwCrs, Result};

/// `secret_coeffs: Vec<Vec<i64>>` (k x l, `src/keys/secret_key.rs:14-18`) as the flat `[k][l]` the ABI takes.
fn flat_secret(sk: &SecretKey) -> Vec<i64> {
    sk.secret_coeffs.iter().flat_map(|row| row.iter().copied()).collect()
}

impl PublicKey {
    /// b = s * A + e (public_key.rs:111-147).  The product and the error sampling run on the device; the row is
    /// written into the resident public-key matrix at `index` and read back for the returned `PublicKey`.
    /// The second element of the reference's result (the error polynomials, kept "for testing") is not produced:
    /// the errors never leave the device and are wiped with the rest of the key material.
    pub fn generate_at<R: RngCore + CryptoRng>(secret_key: &SecretKey, crs: &PvwCrs, index: usize, rng: &mut R) -> Result<Self> {
        if secret_key.params.k != crs.params.k {
            return Err(PvwError::DimensionMismatch { expected: crs.params.k, actual: secret_key.params.k });
        }
        let params = &crs.params;
        let ctx = params.hip.raw();
        let mut sk = flat_secret(secret_key);
        let seed = seed_from_rng(rng);
        let rc = unsafe { sys::pvw_keygen(ctx, index as u32, index as u32 + 1, sk.as_ptr(), std::ptr::null(), seed.as_ptr()) };
        sk.zeroize();
        check(rc)?;
        let words = poly_words(params);
        let mut flat = vec![0u64; params.k * words];
        check(unsafe { sys::pvw_get_pk(ctx, index as u32, index as u32 + 1, flat.as_mut_ptr(), sys::PVW_REPR_POWER) })?;
        let key_polynomials: Result<Vec<Poly>> = flat.chunks_exact(words).map(|c| poly_from_flat(c, params)).collect();
        Ok(Self { key_polynomials: key_polynomials?, params: params.clone() })
    }
}

impl GlobalPublicKey {
    /// `add_public_key` (public_key.rs:214-250): a key generated elsewhere becomes row `index` of the resident matrix.
    pub fn add_public_key(&mut self, index: usize, public_key: PublicKey) -> Result<()> {
        if index >= self.params.n {
            return Err(PvwError::InvalidParameters(format!("Party index {} exceeds maximum {}", index, self.params.n - 1)));
        }
        public_key.validate()?;
        if public_key.params.k != self.params.k {
            return Err(PvwError::InvalidParameters(format!(
                "Public key dimension {} doesn't match global key dimension {}",
                public_key.params.k, self.params.k
            )));
        }
        let mut flat = Vec::with_capacity(self.params.k * poly_words(&self.params));
        for poly in public_key.key_polynomials.iter() {
            poly_to_flat(poly, &mut flat);
        }
        check(unsafe { sys::pvw_load_pk(self.params.hip.raw(), index as u32, index as u32 + 1, flat.as_ptr(), sys::PVW_REPR_POWER) })?;
        for (j, poly) in public_key.key_polynomials.into_iter().enumerate() {
            self.matrix[(index, j)] = poly; // host mirror for get_polynomial / serde
        }
        if index >= self.num_keys {
            self.num_keys = index + 1;
        }
        Ok(())
    }

    /// `generate_and_add_party` (public_key.rs:256-263).
    pub fn generate_and_add_party<R: RngCore + CryptoRng>(&mut self, party: &Party, rng: &mut R) -> Result<()> {
        let public_key = PublicKey::generate_at(party.secret_key(), &self.crs, party.index(), rng)?;
        for (j, poly) in public_key.key_polynomials.into_iter().enumerate() {
            self.matrix[(party.index(), j)] = poly;
        }
        if party.index() >= self.num_keys {
            self.num_keys = party.index() + 1;
        }
        Ok(())
    }

    /// `generate_all_party_keys` (public_key.rs:376-401): the reference generates in parallel (rayon) and adds in
    /// order; here every run of consecutive party indices is ONE batched device call (a modular GEMM per
    /// (limb, slot) on the matrix cores from 8 parties up).
    pub fn generate_all_party_keys(&mut self, parties: &[Party]) -> Result<()> {
        if parties.len() > self.params.n {
            return Err(PvwError::InvalidParameters(format!("Too many parties: {} > {}", parties.len(), self.params.n)));
        }
        let ctx = self.params.hip.raw();
        let seed = seed_from_rng(&mut rand::thread_rng());
        let mut i = 0;
        while i < parties.len() {
            let mut j = i + 1;
            while j < parties.len() && parties[j].index() == parties[j - 1].index() + 1 {
                j += 1;
            }
            let (lo, hi) = (parties[i].index(), parties[i].index() + (j - i));
            if hi > self.params.n {
                return Err(PvwError::InvalidParameters(format!("Party index {} exceeds maximum {}", hi - 1, self.params.n - 1)));
            }
            let mut sk: Vec<i64> = parties[i..j].iter().flat_map(|p| flat_secret(p.secret_key())).collect();
            let rc = unsafe { sys::pvw_keygen(ctx, lo as u32, hi as u32, sk.as_ptr(), std::ptr::null(), seed.as_ptr()) };
            sk.zeroize();
            check(rc)?;
            if hi > self.num_keys {
                self.num_keys = hi;
            }
            i = j;
        }
        self.refresh_host_mirror()?;
        self.prepare_device().map(|_| ())
    }

    /// Build the device-side copies the encrypt calls stream from (`pvw_prepare`: the bit-packed copy of A-hat / B-hat
    /// for `encrypt`, the MFMA-tiled copy for `encrypt_all_party_shares`) NOW, so that the first encrypt after a key
    /// change neither allocates nor waits.  `&self`: the mutators above take `&mut self` (public_key.rs:214-263), so no
    /// encrypt can be in flight while the matrices change; call this after the last `add_public_key`.
    /// Returns the bytes allocated for the copies.
    pub fn prepare_device(&self) -> Result<u64> {
        let mut taken = 0u64;
        check(unsafe {
            sys::pvw_prepare(self.params.hip.raw(), sys::PVW_PREPARE_PACKED | sys::PVW_PREPARE_MFMA, std::ptr::null_mut(), &mut taken)
        })?;
        Ok(taken)
    }

    /// `is_full` (public_key.rs:349-351) as the device sees it.
    pub fn is_full_on_device(&self) -> Result<bool> {
        let mut v = 0i32;
        check(unsafe { sys::pvw_is_full(self.params.hip.raw(), &mut v) })?;
        Ok(v != 0)
    }

    /// Re-read `matrix` from the device (after a batched key generation).
    fn refresh_host_mirror(&mut self) -> Result<()> {
        let (n, k, words) = (self.params.n, self.params.k, poly_words(&self.params));
        let mut flat = vec![0u64; self.num_keys.min(n) * k * words];
        check(unsafe { sys::pvw_get_pk(self.params.hip.raw(), 0, self.num_keys.min(n) as u32, flat.as_mut_ptr(), sys::PVW_REPR_POWER) })?;
        for (idx, chunk) in flat.chunks_exact(words).enumerate() {
            self.matrix[(idx / k, idx % k)] = poly_from_flat(chunk, &self.params)?;
        }
        Ok(())
    }
}
