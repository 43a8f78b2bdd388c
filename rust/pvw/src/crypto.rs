//! `encrypt` / `decrypt_*` over `libpvw_hip.so` (NOT COMPILED here -- see rust/README.md).
//! Replaces the bodies of `src/crypto/encryption.rs:105-296` and `src/crypto/decryption.rs:249-325`; names,
//! signatures, validation order and messages are the reference's.
use std::sync::Arc;

use fhe_math::rq::Poly;
use pvw_hip_sys as sys;
use rand::RngCore;
use zeroize::Zeroize;

use crate::crypto::encryption::PvwCiphertext;
use crate::errors::PvwError;
use crate::ffi_support::{check, poly_from_flat, poly_to_flat, poly_words};
use crate::keys::public_key::GlobalPublicKey;
use crate::keys::secret_key::SecretKey;
use crate::params::{PvwParameters, Result};

fn fresh_seed() -> [u8; 32] {
    let mut seed = [0u8; 32];
    rand::thread_rng().fill_bytes(&mut seed); // fresh randomness per call, as the reference's thread_rng() draws
    seed
}

fn ciphertext_from_flat(c1: &[u64], c2: &[u64], params: &Arc<PvwParameters>) -> Result<PvwCiphertext> {
    let words = poly_words(params);
    let c1: Result<Vec<Poly>> = c1.chunks_exact(words).map(|c| poly_from_flat(c, params)).collect();
    let c2: Result<Vec<Poly>> = c2.chunks_exact(words).map(|c| poly_from_flat(c, params)).collect();
    let ct = PvwCiphertext { c1: c1?, c2: c2?, params: params.clone() };
    ct.validate()?; // encryption.rs:204-211
    Ok(ct)
}

/// `encrypt` (encryption.rs:105-214).  The checks at :109 (scalar count), :117 (key fullness) and :124
/// (correctness condition) run inside `pvw_encrypt` with the reference's messages.
pub fn encrypt(scalars: &[u64], global_pk: &GlobalPublicKey) -> Result<PvwCiphertext> {
    let params = &global_pk.params;
    let words = poly_words(params);
    let rnd = sys::PvwRandomnessT {
        mode: sys::PVW_RND_SEED,
        seed: fresh_seed(),
        r: std::ptr::null(),
        e1: std::ptr::null(),
        e2: std::ptr::null(),
    };
    let (mut c1, mut c2) = (vec![0u64; params.k * words], vec![0u64; params.n * words]);
    check(unsafe {
        sys::pvw_encrypt(params.hip.raw(), scalars.as_ptr(), scalars.len(), &rnd, c1.as_mut_ptr(), c2.as_mut_ptr(), sys::PVW_REPR_POWER)
    })?;
    ciphertext_from_flat(&c1, &c2, params)
}

/// `encrypt_party_shares` (encryption.rs:221-245).
pub fn encrypt_party_shares(party_shares: &[u64], party_index: usize, global_pk: &GlobalPublicKey) -> Result<PvwCiphertext> {
    if party_index >= global_pk.params.n {
        return Err(PvwError::InvalidParameters(format!("Party index {} exceeds maximum {}", party_index, global_pk.params.n - 1)));
    }
    if party_shares.len() != global_pk.params.n {
        return Err(PvwError::InvalidParameters(format!("Party must provide {} shares, got {}", global_pk.params.n, party_shares.len())));
    }
    encrypt(party_shares, global_pk)
}

/// `encrypt_all_party_shares` (encryption.rs:253-286): ONE device call for all dealers instead of a rayon loop
/// over `encrypt` -- the dealers share passes over the resident public key (matrix cores from 3 dealers up), each
/// with its own seed.
pub fn encrypt_all_party_shares(all_shares: &[Vec<u64>], global_pk: &GlobalPublicKey) -> Result<Vec<PvwCiphertext>> {
    let params = &global_pk.params;
    let n = params.n;
    if all_shares.len() != n {
        return Err(PvwError::InvalidParameters(format!("Must provide shares for all {n} parties")));
    }
    for (dealer_idx, dealer_shares) in all_shares.iter().enumerate() {
        if dealer_shares.len() != n {
            return Err(PvwError::InvalidParameters(format!(
                "Dealer {} provided {} shares but needs {}",
                dealer_idx,
                dealer_shares.len(),
                n
            )));
        }
    }
    let words = poly_words(params);
    let scalars: Vec<u64> = all_shares.iter().flat_map(|row| row.iter().copied()).collect();
    let mut seeds = vec![0u8; 32 * n];
    rand::thread_rng().fill_bytes(&mut seeds);
    let (mut c1, mut c2) = (vec![0u64; n * params.k * words], vec![0u64; n * n * words]);
    check(unsafe {
        sys::pvw_encrypt_multi(params.hip.raw(), scalars.as_ptr(), n, n, seeds.as_ptr(), c1.as_mut_ptr(), c2.as_mut_ptr(), sys::PVW_REPR_POWER)
    })?;
    (0..n)
        .map(|d| ciphertext_from_flat(&c1[d * params.k * words..(d + 1) * params.k * words], &c2[d * n * words..(d + 1) * n * words], params))
        .collect()
}

/// `encrypt_broadcast` (encryption.rs:292-296).
pub fn encrypt_broadcast(scalar: u64, global_pk: &GlobalPublicKey) -> Result<PvwCiphertext> {
    let broadcast_values = vec![scalar; global_pk.params.n];
    encrypt(&broadcast_values, global_pk)
}

fn flat_secret(sk: &SecretKey) -> Vec<i64> {
    sk.secret_coeffs.iter().flat_map(|row| row.iter().copied()).collect()
}

/// A `SecretKey` kept on the device in the form the inner products of decrypt read (`pvw_sk_load`: NTT(sk[j]), what
/// `SecretKey::get_polynomial` computes k times per `decrypt_party_value`, secret_key.rs:98-112 / decryption.rs:260).  For a
/// receiver that decrypts many device-resident ciphertext batches under one key: `pvw_decrypt_batch_device_sk` then neither
/// transforms nor wipes per call.  Dropping the handle clears the device copy (`pvw_sk_free`), as dropping the reference's
/// `SecretKey` zeroizes it (secret_key.rs:20-30).  Must not outlive the `PvwParameters` it was loaded for.
pub struct DeviceSecretKey {
    raw: *mut sys::PvwSk,
}

impl DeviceSecretKey {
    pub fn load(secret_key: &SecretKey) -> Result<Self> {
        let mut sk = flat_secret(secret_key);
        let mut raw: *mut sys::PvwSk = std::ptr::null_mut();
        let rc = unsafe { sys::pvw_sk_load(secret_key.params.hip.raw(), sk.as_ptr(), &mut raw) };
        sk.zeroize();
        check(rc)?;
        Ok(Self { raw })
    }

    pub fn raw(&self) -> *const sys::PvwSk {
        self.raw
    }
}

impl Drop for DeviceSecretKey {
    fn drop(&mut self) {
        if !self.raw.is_null() {
            unsafe { sys::pvw_sk_free(self.raw) };
            self.raw = std::ptr::null_mut();
        }
    }
}

/// `decrypt_party_value` (decryption.rs:249-278): <sk, c1> - c2[party_index], inverse NTT and the gadget decode
/// (`decode_scalar_pvw_rns`, :10-58) all on the device; one u64 comes back.
pub fn decrypt_party_value(ciphertext: &PvwCiphertext, secret_key: &SecretKey, party_index: usize) -> Result<u64> {
    let params = &ciphertext.params;
    let mut c1s = Vec::with_capacity(params.k * poly_words(params));
    for poly in ciphertext.c1.iter() {
        poly_to_flat(poly, &mut c1s);
    }
    let mut c2col = Vec::with_capacity(poly_words(params));
    poly_to_flat(&ciphertext.c2[party_index], &mut c2col);
    let mut sk = flat_secret(secret_key);
    let mut out = 0u64;
    let rc = unsafe {
        sys::pvw_decrypt_batch(params.hip.raw(), sk.as_ptr(), c1s.as_ptr(), c2col.as_ptr(), 1, sys::PVW_REPR_POWER, &mut out, std::ptr::null_mut())
    };
    sk.zeroize();
    check(rc)?;
    Ok(out)
}

/// `decrypt_party_shares` (decryption.rs:281-325): one batched device pass over all dealers' ciphertexts.
pub fn decrypt_party_shares(all_ciphertexts: &[PvwCiphertext], secret_key: &SecretKey, party_index: usize) -> Result<Vec<u64>> {
    if all_ciphertexts.is_empty() {
        return Err(PvwError::InvalidParameters("No ciphertexts provided".to_string()));
    }
    let params = &all_ciphertexts[0].params;
    if all_ciphertexts.len() != params.n {
        return Err(PvwError::InvalidParameters(format!("Expected {} ciphertexts, got {}", params.n, all_ciphertexts.len())));
    }
    if party_index >= params.n {
        return Err(PvwError::InvalidParameters(format!("Party index {} exceeds maximum {}", party_index, params.n - 1)));
    }
    let words = poly_words(params);
    let d = all_ciphertexts.len();
    let (mut c1s, mut c2col) = (Vec::with_capacity(d * params.k * words), Vec::with_capacity(d * words));
    for (dealer_idx, ciphertext) in all_ciphertexts.iter().enumerate() {
        ciphertext
            .validate()
            .map_err(|e| PvwError::InvalidParameters(format!("Ciphertext {dealer_idx} invalid: {e}")))?;
        for poly in ciphertext.c1.iter() {
            poly_to_flat(poly, &mut c1s);
        }
        poly_to_flat(&ciphertext.c2[party_index], &mut c2col);
    }
    let mut sk = flat_secret(secret_key);
    let mut out = vec![0u64; d];
    let rc = unsafe {
        sys::pvw_decrypt_batch(params.hip.raw(), sk.as_ptr(), c1s.as_ptr(), c2col.as_ptr(), d, sys::PVW_REPR_POWER, out.as_mut_ptr(), std::ptr::null_mut())
    };
    sk.zeroize();
    check(rc)?;
    Ok(out)
}
