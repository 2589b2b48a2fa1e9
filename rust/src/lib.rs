//! Drop-in for the hot path of `concrete-ntt` on an AMD MI355X: same module / type / method names,
//! every call forwarded to `libcntt_hip.so` (C ABI: `include/cntt.h`).
//!
//! UNVERIFIED SOURCE: the build image has no Rust toolchain; this file documents the binding, it has
//! not been compiled.  The verified surface is the C ABI and its Python mirror.
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_int, c_void};

pub mod ffi {
    use super::*;
    #[repr(C)] pub struct cntt_plan64 { _p: [u8; 0] }
    #[repr(C)] pub struct cntt_plan32 { _p: [u8; 0] }
    #[repr(C)] pub struct cntt_native { _p: [u8; 0] }
    #[repr(C)] pub struct cntt_product { _p: [u8; 0] }
    pub const CNTT_OK: c_int = 0;
    pub const CNTT_NONE: c_int = 1;
    pub const CNTT_EINVAL: c_int = 2;
    pub const CNTT_ELEN: c_int = 3;
    pub const CNTT_MEM_HOST: c_int = 0;
    pub const CNTT_MEM_DEVICE: c_int = 1;
    extern "C" {
        pub fn cntt_last_error() -> *const c_char;
        pub fn cntt_prime64_plan_new(n: usize, p: u64, out: *mut *mut cntt_plan64) -> c_int;
        pub fn cntt_prime64_plan_clone(p: *const cntt_plan64) -> *mut cntt_plan64;
        pub fn cntt_prime64_plan_free(p: *mut cntt_plan64);
        pub fn cntt_prime64_ntt_size(p: *const cntt_plan64) -> usize;
        pub fn cntt_prime64_modulus(p: *const cntt_plan64) -> u64;
        pub fn cntt_prime64_fwd(p: *const cntt_plan64, buf: *mut u64, len: usize) -> c_int;
        pub fn cntt_prime64_inv(p: *const cntt_plan64, buf: *mut u64, len: usize) -> c_int;
        pub fn cntt_prime64_mul_assign_normalize(p: *const cntt_plan64, lhs: *mut u64, ll: usize, rhs: *const u64, rl: usize) -> c_int;
        pub fn cntt_prime64_normalize(p: *const cntt_plan64, v: *mut u64, len: usize) -> c_int;
        pub fn cntt_prime64_mul_accumulate(p: *const cntt_plan64, acc: *mut u64, al: usize, lhs: *const u64, ll: usize, rhs: *const u64, rl: usize) -> c_int;
        pub fn cntt_prime64_fwd_batch(p: *const cntt_plan64, bufs: *mut u64, batch: usize, mem: c_int, stream: *mut c_void) -> c_int;
        pub fn cntt_prime64_inv_batch(p: *const cntt_plan64, bufs: *mut u64, batch: usize, mem: c_int, stream: *mut c_void) -> c_int;
        pub fn cntt_prime64_mul_assign_normalize_batch(p: *const cntt_plan64, lhs: *mut u64, rhs: *const u64, batch: usize, mem: c_int, stream: *mut c_void) -> c_int;

        pub fn cntt_prime32_plan_new(n: usize, p: u32, out: *mut *mut cntt_plan32) -> c_int;
        pub fn cntt_prime32_plan_clone(p: *const cntt_plan32) -> *mut cntt_plan32;
        pub fn cntt_prime32_plan_free(p: *mut cntt_plan32);
        pub fn cntt_prime32_ntt_size(p: *const cntt_plan32) -> usize;
        pub fn cntt_prime32_modulus(p: *const cntt_plan32) -> u32;
        pub fn cntt_prime32_fwd(p: *const cntt_plan32, buf: *mut u32, len: usize) -> c_int;
        pub fn cntt_prime32_inv(p: *const cntt_plan32, buf: *mut u32, len: usize) -> c_int;
        pub fn cntt_prime32_mul_assign_normalize(p: *const cntt_plan32, lhs: *mut u32, ll: usize, rhs: *const u32, rl: usize) -> c_int;
        pub fn cntt_prime32_normalize(p: *const cntt_plan32, v: *mut u32, len: usize) -> c_int;
        pub fn cntt_prime32_mul_accumulate(p: *const cntt_plan32, acc: *mut u32, al: usize, lhs: *const u32, ll: usize, rhs: *const u32, rl: usize) -> c_int;

        pub fn cntt_native_plan_new(kind: c_int, n: usize, out: *mut *mut cntt_native) -> c_int;
        pub fn cntt_native_plan_free(p: *mut cntt_native);
        pub fn cntt_native_ntt_size(p: *const cntt_native) -> usize;
        pub fn cntt_native_ntt32(p: *const cntt_native, i: c_int) -> *const cntt_plan32;
        pub fn cntt_native_fwd(p: *const cntt_native, value: *const c_void, len: usize, residues: *const *mut c_void) -> c_int;
        pub fn cntt_native_fwd_binary(p: *const cntt_native, value: *const c_void, len: usize, residues: *const *mut c_void) -> c_int;
        pub fn cntt_native_inv(p: *const cntt_native, value: *mut c_void, len: usize, residues: *const *mut c_void) -> c_int;
        pub fn cntt_native_negacyclic_polymul(p: *const cntt_native, prod: *mut c_void, pl: usize, lhs: *const c_void, ll: usize, rhs: *const c_void, rl: usize) -> c_int;
        pub fn cntt_product_plan_new(n: usize, modulus: u64, factors: *const u64, nfactors: usize, out: *mut *mut cntt_product) -> c_int;
        pub fn cntt_product_plan_clone(p: *const cntt_product) -> *mut cntt_product;
        pub fn cntt_product_plan_free(p: *mut cntt_product);
        pub fn cntt_product_ntt_size(p: *const cntt_product) -> usize;
        pub fn cntt_product_modulus(p: *const cntt_product) -> u64;
        pub fn cntt_product_ntt_domain_len(p: *const cntt_product) -> usize;
        pub fn cntt_product_fwd(p: *const cntt_product, ntt: *mut u64, nl: usize, standard: *const u64, sl: usize, mode: c_int, bound: u64) -> c_int;
        pub fn cntt_product_inv(p: *const cntt_product, standard: *mut u64, sl: usize, ntt: *mut u64, nl: usize, mode: c_int) -> c_int;
        pub fn cntt_product_mul_assign_normalize(p: *const cntt_product, lhs: *mut u64, ll: usize, rhs: *const u64, rl: usize) -> c_int;
        pub fn cntt_product_normalize(p: *const cntt_product, v: *mut u64, len: usize) -> c_int;
        pub fn cntt_product_mul_accumulate(p: *const cntt_product, acc: *mut u64, al: usize, lhs: *const u64, ll: usize, rhs: *const u64, rl: usize) -> c_int;
        pub fn cntt_native_negacyclic_polymul_batch(p: *const cntt_native, prod: *mut c_void, lhs: *const c_void, rhs: *const c_void, batch: usize, mem: c_int, stream: *mut c_void) -> c_int;
    }
}

/// Status -> the reference's behaviour: `CNTT_EINVAL` / `CNTT_ELEN` are the reference's panics.
#[track_caller]
fn check(rc: c_int) {
    if rc != ffi::CNTT_OK {
        let msg = unsafe { core::ffi::CStr::from_ptr(ffi::cntt_last_error()) };
        panic!("concrete-ntt-hip: {}", msg.to_string_lossy());
    }
}

pub mod prime64 {
    use super::*;
    /// Negacyclic NTT plan for 64bit primes (concrete_ntt::prime64::Plan, src/prime64.rs:220-236).
    pub struct Plan(pub(crate) *mut ffi::cntt_plan64);
    unsafe impl Send for Plan {}
    unsafe impl Sync for Plan {}
    impl Plan {
        /// src/prime64.rs:704 -- `None` when the C ABI reports CNTT_NONE; panics where the reference panics.
        pub fn try_new(polynomial_size: usize, modulus: u64) -> Option<Self> {
            let mut out = core::ptr::null_mut();
            match unsafe { ffi::cntt_prime64_plan_new(polynomial_size, modulus, &mut out) } {
                ffi::CNTT_OK => Some(Self(out)),
                ffi::CNTT_NONE => None,
                rc => { check(rc); None }
            }
        }
        pub fn ntt_size(&self) -> usize { unsafe { ffi::cntt_prime64_ntt_size(self.0) } }
        pub fn modulus(&self) -> u64 { unsafe { ffi::cntt_prime64_modulus(self.0) } }
        pub fn fwd(&self, buf: &mut [u64]) { check(unsafe { ffi::cntt_prime64_fwd(self.0, buf.as_mut_ptr(), buf.len()) }) }
        pub fn inv(&self, buf: &mut [u64]) { check(unsafe { ffi::cntt_prime64_inv(self.0, buf.as_mut_ptr(), buf.len()) }) }
        pub fn mul_assign_normalize(&self, lhs: &mut [u64], rhs: &[u64]) {
            check(unsafe { ffi::cntt_prime64_mul_assign_normalize(self.0, lhs.as_mut_ptr(), lhs.len(), rhs.as_ptr(), rhs.len()) })
        }
        pub fn normalize(&self, values: &mut [u64]) { check(unsafe { ffi::cntt_prime64_normalize(self.0, values.as_mut_ptr(), values.len()) }) }
        pub fn mul_accumulate(&self, acc: &mut [u64], lhs: &[u64], rhs: &[u64]) {
            check(unsafe { ffi::cntt_prime64_mul_accumulate(self.0, acc.as_mut_ptr(), acc.len(), lhs.as_ptr(), lhs.len(), rhs.as_ptr(), rhs.len()) })
        }
        /// GPU fast path (not in the reference): `batch` polynomials back to back in host memory.
        pub fn fwd_batch(&self, bufs: &mut [u64]) {
            assert_eq!(bufs.len() % self.ntt_size(), 0);
            check(unsafe { ffi::cntt_prime64_fwd_batch(self.0, bufs.as_mut_ptr(), bufs.len() / self.ntt_size(), ffi::CNTT_MEM_HOST, core::ptr::null_mut()) })
        }
        /// Same on a device pointer + HIP stream owned by the caller.
        /// # Safety: `dev` must address `batch * ntt_size` u64 of device memory valid on `stream`.
        pub unsafe fn fwd_batch_device(&self, dev: *mut u64, batch: usize, stream: *mut c_void) {
            check(ffi::cntt_prime64_fwd_batch(self.0, dev, batch, ffi::CNTT_MEM_DEVICE, stream))
        }
    }
    impl Clone for Plan { fn clone(&self) -> Self { Self(unsafe { ffi::cntt_prime64_plan_clone(self.0) }) } }
    impl Drop for Plan { fn drop(&mut self) { unsafe { ffi::cntt_prime64_plan_free(self.0) } } }
    impl core::fmt::Debug for Plan { // src/prime64.rs:238-245
        fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
            f.debug_struct("Plan").field("ntt_size", &self.ntt_size()).field("modulus", &self.modulus()).finish()
        }
    }
    /// src/prime64/generic_solinas.rs:35-40
    pub struct Solinas;
    impl Solinas { pub const P: u64 = ((1u128 << 64) - (1u128 << 32) + 1u128) as u64; }
}

pub mod prime32 {
    use super::*;
    /// Negacyclic NTT plan for 32bit primes (concrete_ntt::prime32::Plan, src/prime32.rs:600-616).
    pub struct Plan(pub(crate) *mut ffi::cntt_plan32);
    unsafe impl Send for Plan {}
    unsafe impl Sync for Plan {}
    impl Plan {
        pub fn try_new(polynomial_size: usize, modulus: u32) -> Option<Self> {
            let mut out = core::ptr::null_mut();
            match unsafe { ffi::cntt_prime32_plan_new(polynomial_size, modulus, &mut out) } {
                ffi::CNTT_OK => Some(Self(out)),
                ffi::CNTT_NONE => None,
                rc => { check(rc); None }
            }
        }
        pub fn ntt_size(&self) -> usize { unsafe { ffi::cntt_prime32_ntt_size(self.0) } }
        pub fn modulus(&self) -> u32 { unsafe { ffi::cntt_prime32_modulus(self.0) } }
        pub fn fwd(&self, buf: &mut [u32]) { check(unsafe { ffi::cntt_prime32_fwd(self.0, buf.as_mut_ptr(), buf.len()) }) }
        pub fn inv(&self, buf: &mut [u32]) { check(unsafe { ffi::cntt_prime32_inv(self.0, buf.as_mut_ptr(), buf.len()) }) }
        pub fn mul_assign_normalize(&self, lhs: &mut [u32], rhs: &[u32]) {
            check(unsafe { ffi::cntt_prime32_mul_assign_normalize(self.0, lhs.as_mut_ptr(), lhs.len(), rhs.as_ptr(), rhs.len()) })
        }
        pub fn normalize(&self, values: &mut [u32]) { check(unsafe { ffi::cntt_prime32_normalize(self.0, values.as_mut_ptr(), values.len()) }) }
        pub fn mul_accumulate(&self, acc: &mut [u32], lhs: &[u32], rhs: &[u32]) {
            check(unsafe { ffi::cntt_prime32_mul_accumulate(self.0, acc.as_mut_ptr(), acc.len(), lhs.as_ptr(), lhs.len(), rhs.as_ptr(), rhs.len()) })
        }
    }
    impl Clone for Plan { fn clone(&self) -> Self { Self(unsafe { ffi::cntt_prime32_plan_clone(self.0) }) } }
    impl Drop for Plan { fn drop(&mut self) { unsafe { ffi::cntt_prime32_plan_free(self.0) } } }
}

/// native64::Plan32 (src/native64.rs:14-22, :930-1070); the other native plans follow the same
/// pattern with kind = 0 (native32), 2 (native128, u128 words), 3..5 (native_binary*), 6..9 (Plan52).
pub mod native64 {
    use super::*;
    pub struct Plan32(*mut ffi::cntt_native);
    unsafe impl Send for Plan32 {}
    unsafe impl Sync for Plan32 {}
    impl Plan32 {
        pub fn try_new(n: usize) -> Option<Self> {
            let mut out = core::ptr::null_mut();
            match unsafe { ffi::cntt_native_plan_new(1, n, &mut out) } {
                ffi::CNTT_OK => Some(Self(out)),
                ffi::CNTT_NONE => None,
                rc => { check(rc); None }
            }
        }
        pub fn ntt_size(&self) -> usize { unsafe { ffi::cntt_native_ntt_size(self.0) } }
        pub fn fwd(&self, value: &[u64], mod_p0: &mut [u32], mod_p1: &mut [u32], mod_p2: &mut [u32], mod_p3: &mut [u32], mod_p4: &mut [u32]) {
            let r = [mod_p0.as_mut_ptr() as *mut c_void, mod_p1.as_mut_ptr() as _, mod_p2.as_mut_ptr() as _, mod_p3.as_mut_ptr() as _, mod_p4.as_mut_ptr() as _];
            check(unsafe { ffi::cntt_native_fwd(self.0, value.as_ptr() as _, value.len(), r.as_ptr()) })
        }
        pub fn inv(&self, value: &mut [u64], mod_p0: &mut [u32], mod_p1: &mut [u32], mod_p2: &mut [u32], mod_p3: &mut [u32], mod_p4: &mut [u32]) {
            let r = [mod_p0.as_mut_ptr() as *mut c_void, mod_p1.as_mut_ptr() as _, mod_p2.as_mut_ptr() as _, mod_p3.as_mut_ptr() as _, mod_p4.as_mut_ptr() as _];
            check(unsafe { ffi::cntt_native_inv(self.0, value.as_mut_ptr() as _, value.len(), r.as_ptr()) })
        }
        pub fn negacyclic_polymul(&self, prod: &mut [u64], lhs: &[u64], rhs: &[u64]) {
            check(unsafe { ffi::cntt_native_negacyclic_polymul(self.0, prod.as_mut_ptr() as _, prod.len(), lhs.as_ptr() as _, lhs.len(), rhs.as_ptr() as _, rhs.len()) })
        }
    }
    impl Drop for Plan32 { fn drop(&mut self) { unsafe { ffi::cntt_native_plan_free(self.0) } } }
}

/// product::Plan (src/product.rs:139-967).
pub mod product {
    use super::*;
    #[derive(Copy, Clone, Debug)]
    pub enum FwdMode { Generic, Bounded(u64) }
    #[derive(Copy, Clone, Debug)]
    pub enum InvMode { Replace, Accumulate }
    pub struct Plan(*mut ffi::cntt_product);
    unsafe impl Send for Plan {}
    unsafe impl Sync for Plan {}
    impl Plan {
        pub fn try_new(polynomial_size: usize, modulus: u64, factors: impl IntoIterator<Item = u64>) -> Option<Self> {
            let f: Vec<u64> = factors.into_iter().collect();
            let mut out = core::ptr::null_mut();
            match unsafe { ffi::cntt_product_plan_new(polynomial_size, modulus, f.as_ptr(), f.len(), &mut out) } {
                ffi::CNTT_OK => Some(Self(out)),
                ffi::CNTT_NONE => None,
                rc => { check(rc); None }
            }
        }
        pub fn ntt_size(&self) -> usize { unsafe { ffi::cntt_product_ntt_size(self.0) } }
        pub fn modulus(&self) -> u64 { unsafe { ffi::cntt_product_modulus(self.0) } }
        pub fn ntt_domain_len(&self) -> usize { unsafe { ffi::cntt_product_ntt_domain_len(self.0) } }
        #[track_caller]
        pub fn fwd(&self, ntt: &mut [u64], standard: &[u64], mode: FwdMode) {
            let (m, b) = match mode { FwdMode::Generic => (0, 0), FwdMode::Bounded(b) => (1, b) };
            check(unsafe { ffi::cntt_product_fwd(self.0, ntt.as_mut_ptr(), ntt.len(), standard.as_ptr(), standard.len(), m, b) })
        }
        #[track_caller]
        pub fn inv(&self, standard: &mut [u64], ntt: &mut [u64], mode: InvMode) {
            let m = match mode { InvMode::Replace => 0, InvMode::Accumulate => 1 };
            check(unsafe { ffi::cntt_product_inv(self.0, standard.as_mut_ptr(), standard.len(), ntt.as_mut_ptr(), ntt.len(), m) })
        }
        #[track_caller]
        pub fn mul_assign_normalize(&self, lhs: &mut [u64], rhs: &[u64]) {
            check(unsafe { ffi::cntt_product_mul_assign_normalize(self.0, lhs.as_mut_ptr(), lhs.len(), rhs.as_ptr(), rhs.len()) })
        }
        #[track_caller]
        pub fn normalize(&self, values: &mut [u64]) {
            check(unsafe { ffi::cntt_product_normalize(self.0, values.as_mut_ptr(), values.len()) })
        }
        #[track_caller]
        pub fn mul_accumulate(&self, acc: &mut [u64], lhs: &[u64], rhs: &[u64]) {
            check(unsafe { ffi::cntt_product_mul_accumulate(self.0, acc.as_mut_ptr(), acc.len(), lhs.as_ptr(), lhs.len(), rhs.as_ptr(), rhs.len()) })
        }
    }
    impl Clone for Plan { fn clone(&self) -> Self { Self(unsafe { ffi::cntt_product_plan_clone(self.0) }) } }
    impl Drop for Plan { fn drop(&mut self) { unsafe { ffi::cntt_product_plan_free(self.0) } } }
}
