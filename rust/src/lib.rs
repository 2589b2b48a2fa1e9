//! Drop-in for the hot path of `concrete-ntt` (v0.2.0) on an AMD MI355X: the crate's module / type / method names
//! (`src/lib.rs:88-112` of the reference), every call forwarded to `libcntt_hip.so` through the C ABI of
//! `include/cntt.h`.  `ffi.rs` is generated from that header (`tools/gen_rust_ffi.py`); `tests/test_rust_shim.py`
//! re-parses header and binding independently and compares all 84 entry points, and checks that every public item of
//! the reference's twelve modules exists here with the reference's signature.
//!
//! NOT COMPILED: the build image has no Rust toolchain.  The verified surfaces are the C ABI, its C++
//! mirror (`include/cntt.hpp`) and its Python mirror; this crate is the binding a maintainer adds on top.
//!
//! Reference behaviour kept: `try_new -> Option` (None = `CNTT_NONE`), contract violations panic (`CNTT_EINVAL`,
//! `CNTT_ELEN`), plans are `Send + Sync + Clone` (+ `Debug` where the reference derives it), host slices in place.
//! Added (not in the reference): `*_batch` methods on host slices and `*_batch_device` on raw device pointers + a HIP
//! stream -- the path that shows GPU speed.
#![allow(non_camel_case_types, clippy::too_many_arguments, clippy::missing_safety_doc)]

use core::ffi::{c_int, c_void};

pub mod ffi;

/// HIP stream handle (`hipStream_t`); null = the default stream.
pub type Stream = *mut c_void;

#[track_caller]
fn check(rc: c_int) {
    if rc != ffi::CNTT_OK {
        let msg = unsafe { core::ffi::CStr::from_ptr(ffi::cntt_last_error()) };
        panic!("concrete-ntt-hip: {}", msg.to_string_lossy());
    }
}

/// `Option` protocol of every `try_new`: `CNTT_NONE` is the reference's `None`, anything else its panic.
#[track_caller]
fn option_of<T>(rc: c_int, handle: *mut T) -> Option<*mut T> {
    match rc {
        ffi::CNTT_OK => Some(handle),
        ffi::CNTT_NONE => None,
        rc => {
            check(rc);
            None
        }
    }
}

/// Number of visible HIP devices (0: no GPU / no driver -- every transform then panics: there is no CPU path).
pub fn device_count() -> usize {
    unsafe { ffi::cntt_device_count() as usize }
}

/// `"cntt-hip <version> (gfx950) csrc:<hash of the kernel sources>"`.
pub fn version() -> String {
    unsafe { core::ffi::CStr::from_ptr(ffi::cntt_version()) }.to_string_lossy().into_owned()
}

/// `[begin, end)` of `rank`'s contiguous shard of a batch of independent polynomials split over `world` devices (remainders go to
/// the low ranks).  Pure arithmetic; the caller drives the devices (one thread + one stream per device, `examples/multi_device.cpp`).
pub fn shard_bounds(batch: usize, world: usize, rank: usize) -> core::ops::Range<usize> {
    let (mut b, mut e) = (0usize, 0usize);
    check(unsafe { ffi::cntt_shard_bounds(batch, world as c_int, rank as c_int, &mut b, &mut e) });
    b..e
}

/// TESTING ONLY (`include/cntt.h`): kernel-selection switch `key` := `value` (`-1`: its default; key `"reset"`: all defaults).
/// Results are identical for every setting; the library never reads the process environment.
pub fn debug_set(key: &str, value: i32) {
    let k = std::ffi::CString::new(key).expect("switch names hold no NUL");
    check(unsafe { ffi::cntt_debug_set(k.as_ptr(), value as c_int) })
}
pub fn debug_get(key: &str) -> i32 {
    let k = std::ffi::CString::new(key).expect("switch names hold no NUL");
    let mut v: c_int = 0;
    check(unsafe { ffi::cntt_debug_get(k.as_ptr(), &mut v) });
    v as i32
}

/// Synthetic inputs generated on the device; `dev` is device memory.
pub unsafe fn fill_uniform_u64(dev: *mut u64, count: usize, bound: u64, seed: u64, stream: Stream) {
    check(ffi::cntt_fill_uniform_u64(dev, count, bound, seed, stream))
}
pub unsafe fn fill_uniform_u32(dev: *mut u32, count: usize, bound: u32, seed: u64, stream: Stream) {
    check(ffi::cntt_fill_uniform_u32(dev, count, bound, seed, stream))
}

/// Scalar fields of a prime plan (private in the reference; exposed for parity tests).
pub type PlanInfo = ffi::cntt_plan_info;

// -------------------------------------------------------------------------------------------------------------------
// prime32::Plan / prime64::Plan  (src/prime32.rs:601-927, src/prime64.rs:221-1128)
// -------------------------------------------------------------------------------------------------------------------
macro_rules! prime_plan {
    ($word:ty, $handle:ident, $new:ident, $clone:ident, $free:ident, $size:ident, $modulus:ident, $info:ident, $table:ident,
     $fwd:ident, $inv:ident, $mul:ident, $norm:ident, $acc:ident,
     $fwd_b:ident, $inv_b:ident, $mul_b:ident, $norm_b:ident, $acc_b:ident, $mulntt_b:ident, $ext_b:ident) => {
        /// Negacyclic NTT plan for a prime modulus (same type name and methods as the reference's `Plan`).
        pub struct Plan {
            raw: *mut ffi::$handle,
            owned: bool, // false: borrowed from a native / product plan (`ntt_0()` ...), never freed
        }
        unsafe impl Send for Plan {}
        unsafe impl Sync for Plan {}
        impl Plan {
            /// `Plan::try_new(polynomial_size, modulus)`: `None` if the size is not a power of two (or too small), the
            /// modulus is not prime, or no primitive 2n-th root exists; panics on modulus <= 1 like `Div::new`.
            #[track_caller]
            pub fn try_new(polynomial_size: usize, modulus: $word) -> Option<Self> {
                let mut out = core::ptr::null_mut();
                let rc = unsafe { ffi::$new(polynomial_size, modulus, &mut out) };
                option_of(rc, out).map(|raw| Self { raw, owned: true })
            }
            /// A plan that lives inside a native / product plan: handed out only as a `PlanRef<'parent>` (never dropped, never
            /// outliving the parent: the C++ object belongs to the parent handle and is freed with it).
            pub(crate) unsafe fn borrowed<'a>(raw: *const ffi::$handle) -> PlanRef<'a> {
                PlanRef { plan: core::mem::ManuallyDrop::new(Self { raw: raw as *mut _, owned: false }), _parent: core::marker::PhantomData }
            }
            pub fn ntt_size(&self) -> usize { unsafe { ffi::$size(self.raw) } }
            pub fn modulus(&self) -> $word { unsafe { ffi::$modulus(self.raw) } }
            /// p_barrett, big_q, n_inv_mod_p, root ... (private fields of the reference)
            pub fn info(&self) -> PlanInfo {
                let mut i = PlanInfo::default();
                check(unsafe { ffi::$info(self.raw, &mut i) });
                i
            }
            /// Table `which` (0 twid, 1 twid_shoup, 2 inv_twid, 3 inv_twid_shoup); `None` if the plan has no Shoup tables.
            pub fn table(&self, which: usize) -> Option<Vec<$word>> {
                let mut t = vec![0 as $word; self.ntt_size()];
                match unsafe { ffi::$table(self.raw, which as c_int, t.as_mut_ptr(), t.len()) } {
                    ffi::CNTT_NONE => None,
                    rc => { check(rc); Some(t) }
                }
            }
            /// In-place forward transform, standard order in, bit-reversed order out; panics unless `buf.len() == ntt_size()`.
            /// Input contract (as the reference's tests feed it, include/cntt.h): every coefficient `< modulus`.  Larger words are not
            /// rejected and not reduced: the call completes and the outputs they reach are unspecified residues.
            #[track_caller]
            pub fn fwd(&self, buf: &mut [$word]) { check(unsafe { ffi::$fwd(self.raw, buf.as_mut_ptr(), buf.len()) }) }
            /// In-place inverse transform (unnormalised: `inv(fwd(x)) = n x`).
            #[track_caller]
            pub fn inv(&self, buf: &mut [$word]) { check(unsafe { ffi::$inv(self.raw, buf.as_mut_ptr(), buf.len()) }) }
            #[track_caller]
            pub fn mul_assign_normalize(&self, lhs: &mut [$word], rhs: &[$word]) {
                check(unsafe { ffi::$mul(self.raw, lhs.as_mut_ptr(), lhs.len(), rhs.as_ptr(), rhs.len()) })
            }
            #[track_caller]
            pub fn normalize(&self, values: &mut [$word]) { check(unsafe { ffi::$norm(self.raw, values.as_mut_ptr(), values.len()) }) }
            #[track_caller]
            pub fn mul_accumulate(&self, acc: &mut [$word], lhs: &[$word], rhs: &[$word]) {
                check(unsafe { ffi::$acc(self.raw, acc.as_mut_ptr(), acc.len(), lhs.as_ptr(), lhs.len(), rhs.as_ptr(), rhs.len()) })
            }

            // ---- batched host slices: `batch` polynomials back to back (staged through the device) ----
            #[track_caller]
            fn batch_of(&self, len: usize) -> usize {
                assert_eq!(len % self.ntt_size(), 0, "a batch is whole polynomials");
                len / self.ntt_size()
            }
            #[track_caller]
            pub fn fwd_batch(&self, bufs: &mut [$word]) {
                let b = self.batch_of(bufs.len());
                check(unsafe { ffi::$fwd_b(self.raw, bufs.as_mut_ptr(), b, ffi::CNTT_MEM_HOST, core::ptr::null_mut()) })
            }
            #[track_caller]
            pub fn inv_batch(&self, bufs: &mut [$word]) {
                let b = self.batch_of(bufs.len());
                check(unsafe { ffi::$inv_b(self.raw, bufs.as_mut_ptr(), b, ffi::CNTT_MEM_HOST, core::ptr::null_mut()) })
            }
            #[track_caller]
            pub fn mul_assign_normalize_batch(&self, lhs: &mut [$word], rhs: &[$word]) {
                assert_eq!(lhs.len(), rhs.len());
                let b = self.batch_of(lhs.len());
                check(unsafe { ffi::$mul_b(self.raw, lhs.as_mut_ptr(), rhs.as_ptr(), b, ffi::CNTT_MEM_HOST, core::ptr::null_mut()) })
            }
            #[track_caller]
            pub fn normalize_batch(&self, values: &mut [$word]) {
                let b = self.batch_of(values.len());
                check(unsafe { ffi::$norm_b(self.raw, values.as_mut_ptr(), b, ffi::CNTT_MEM_HOST, core::ptr::null_mut()) })
            }
            #[track_caller]
            pub fn mul_accumulate_batch(&self, acc: &mut [$word], lhs: &[$word], rhs: &[$word]) {
                assert!(acc.len() == lhs.len() && lhs.len() == rhs.len());
                let b = self.batch_of(acc.len());
                check(unsafe { ffi::$acc_b(self.raw, acc.as_mut_ptr(), lhs.as_ptr(), rhs.as_ptr(), b, ffi::CNTT_MEM_HOST, core::ptr::null_mut()) })
            }
            /// `lhs <- inv(mul_assign_normalize(fwd(lhs), rhs_ntt))` per polynomial, one fused kernel where the size allows.
            #[track_caller]
            pub fn mul_ntt_batch(&self, lhs: &mut [$word], rhs_ntt: &[$word]) {
                assert_eq!(lhs.len(), rhs_ntt.len());
                let b = self.batch_of(lhs.len());
                check(unsafe { ffi::$mulntt_b(self.raw, lhs.as_mut_ptr(), rhs_ntt.as_ptr(), b, ffi::CNTT_MEM_HOST, core::ptr::null_mut()) })
            }
            /// `out[b][o] (+)= inv(sum_j fwd(terms[b][j]) . key_ntt[j][o])`: the fused `mul_accumulate` chain.
            #[track_caller]
            pub fn external_product_batch(&self, out: &mut [$word], terms: &[$word], key_ntt: &[$word], nterms: usize, nout: usize, accumulate: bool) {
                let n = self.ntt_size();
                assert!(nout > 0 && out.len() % (n * nout) == 0);
                let b = out.len() / (n * nout);
                assert!(terms.len() == b * nterms * n && key_ntt.len() == nterms * nout * n);
                check(unsafe { ffi::$ext_b(self.raw, out.as_mut_ptr(), terms.as_ptr(), key_ntt.as_ptr(), nterms, nout, b, accumulate as c_int, ffi::CNTT_MEM_HOST, core::ptr::null_mut()) })
            }

            // ---- device-resident batches: raw device pointers + a HIP stream; enqueue only, never synchronise ----
            pub unsafe fn fwd_batch_device(&self, dev: *mut $word, batch: usize, stream: Stream) {
                check(ffi::$fwd_b(self.raw, dev, batch, ffi::CNTT_MEM_DEVICE, stream))
            }
            pub unsafe fn inv_batch_device(&self, dev: *mut $word, batch: usize, stream: Stream) {
                check(ffi::$inv_b(self.raw, dev, batch, ffi::CNTT_MEM_DEVICE, stream))
            }
            pub unsafe fn mul_assign_normalize_batch_device(&self, lhs: *mut $word, rhs: *const $word, batch: usize, stream: Stream) {
                check(ffi::$mul_b(self.raw, lhs, rhs, batch, ffi::CNTT_MEM_DEVICE, stream))
            }
            pub unsafe fn normalize_batch_device(&self, values: *mut $word, batch: usize, stream: Stream) {
                check(ffi::$norm_b(self.raw, values, batch, ffi::CNTT_MEM_DEVICE, stream))
            }
            pub unsafe fn mul_accumulate_batch_device(&self, acc: *mut $word, lhs: *const $word, rhs: *const $word, batch: usize, stream: Stream) {
                check(ffi::$acc_b(self.raw, acc, lhs, rhs, batch, ffi::CNTT_MEM_DEVICE, stream))
            }
            pub unsafe fn mul_ntt_batch_device(&self, lhs: *mut $word, rhs_ntt: *const $word, batch: usize, stream: Stream) {
                check(ffi::$mulntt_b(self.raw, lhs, rhs_ntt, batch, ffi::CNTT_MEM_DEVICE, stream))
            }
            pub unsafe fn external_product_batch_device(&self, out: *mut $word, terms: *const $word, key_ntt: *const $word, nterms: usize, nout: usize, batch: usize, accumulate: bool, stream: Stream) {
                check(ffi::$ext_b(self.raw, out, terms, key_ntt, nterms, nout, batch, accumulate as c_int, ffi::CNTT_MEM_DEVICE, stream))
            }
        }
        impl Clone for Plan {
            fn clone(&self) -> Self { Self { raw: unsafe { ffi::$clone(self.raw) }, owned: true } }
        }
        impl Drop for Plan {
            fn drop(&mut self) { if self.owned { unsafe { ffi::$free(self.raw) } } }
        }
        /// What `ntt_0()` ... of the native plans and `plan_32()` / `plan_64()` of the product plan return where the reference returns
        /// `&Plan`: a borrow-carrying handle that derefs to `Plan` (every `&self` method; `.clone()` gives an owned, independent
        /// plan).  The lifetime is the borrow of the parent plan, so safe code cannot use it after the parent is dropped.
        pub struct PlanRef<'a> {
            plan: core::mem::ManuallyDrop<Plan>,
            _parent: core::marker::PhantomData<&'a ()>,
        }
        impl<'a> core::ops::Deref for PlanRef<'a> {
            type Target = Plan;
            fn deref(&self) -> &Plan { &self.plan }
        }
        impl<'a> core::fmt::Debug for PlanRef<'a> {
            fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result { core::fmt::Debug::fmt(&**self, f) }
        }
        /// Like the reference (src/prime64.rs:238-245): only `ntt_size` and `modulus`.
        impl core::fmt::Debug for Plan {
            fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
                f.debug_struct("Plan").field("ntt_size", &self.ntt_size()).field("modulus", &self.modulus()).finish()
            }
        }
    };
}

/// 32bit negacyclic NTT for a prime modulus.
pub mod prime32 {
    use super::*;
    prime_plan!(u32, cntt_plan32, cntt_prime32_plan_new, cntt_prime32_plan_clone, cntt_prime32_plan_free, cntt_prime32_ntt_size,
                cntt_prime32_modulus, cntt_prime32_plan_info, cntt_prime32_plan_table, cntt_prime32_fwd, cntt_prime32_inv,
                cntt_prime32_mul_assign_normalize, cntt_prime32_normalize, cntt_prime32_mul_accumulate, cntt_prime32_fwd_batch,
                cntt_prime32_inv_batch, cntt_prime32_mul_assign_normalize_batch, cntt_prime32_normalize_batch,
                cntt_prime32_mul_accumulate_batch, cntt_prime32_mul_ntt_batch, cntt_prime32_external_product_batch);
}

/// 64bit negacyclic NTT for a prime modulus.
pub mod prime64 {
    use super::*;
    prime_plan!(u64, cntt_plan64, cntt_prime64_plan_new, cntt_prime64_plan_clone, cntt_prime64_plan_free, cntt_prime64_ntt_size,
                cntt_prime64_modulus, cntt_prime64_plan_info, cntt_prime64_plan_table, cntt_prime64_fwd, cntt_prime64_inv,
                cntt_prime64_mul_assign_normalize, cntt_prime64_normalize, cntt_prime64_mul_accumulate, cntt_prime64_fwd_batch,
                cntt_prime64_inv_batch, cntt_prime64_mul_assign_normalize_batch, cntt_prime64_normalize_batch,
                cntt_prime64_mul_accumulate_batch, cntt_prime64_mul_ntt_batch, cntt_prime64_external_product_batch);
    /// src/prime64/generic_solinas.rs:35-40
    #[derive(Copy, Clone, Debug)]
    pub struct Solinas;
    impl Solinas {
        pub const P: u64 = ((1u128 << 64) - (1u128 << 32) + 1u128) as u64;
    }
}

// -------------------------------------------------------------------------------------------------------------------
// native* / native_binary* plans.  One C handle type; `kind` = cntt_native_kind_t selects the reference type.
// The residue buffers are the reference's separate `mod_p0 ..` arguments.
// -------------------------------------------------------------------------------------------------------------------
/// Shared implementation behind every native plan type.
pub struct NativeRaw(*mut ffi::cntt_native);
unsafe impl Send for NativeRaw {}
unsafe impl Sync for NativeRaw {}
impl NativeRaw {
    #[track_caller]
    fn try_new(kind: c_int, n: usize) -> Option<Self> {
        let mut out = core::ptr::null_mut();
        let rc = unsafe { ffi::cntt_native_plan_new(kind, n, &mut out) };
        option_of(rc, out).map(Self)
    }
    fn ntt_size(&self) -> usize { unsafe { ffi::cntt_native_ntt_size(self.0) } }
    fn nprimes(&self) -> usize { unsafe { ffi::cntt_native_nprimes(self.0) as usize } }
    fn word_bytes(&self) -> usize { unsafe { ffi::cntt_native_word_bytes(self.0) as usize } }
    fn residue_bytes(&self) -> usize { unsafe { ffi::cntt_native_residue_bytes(self.0) as usize } }
    fn reserve(&self, batch: usize) { check(unsafe { ffi::cntt_native_reserve(self.0, batch) }) }
}
impl Clone for NativeRaw {
    fn clone(&self) -> Self { Self(unsafe { ffi::cntt_native_plan_clone(self.0) }) }
}
impl Drop for NativeRaw {
    fn drop(&mut self) { unsafe { ffi::cntt_native_plan_free(self.0) } }
}

macro_rules! native_plan {
    // $name: type name; $kind: cntt_native_kind_t; $w: coefficient word; $r: residue word; $k: number of primes;
    // $sub: prime module of the sub-plans; $getter: cntt_native_ntt32 / ntt64; [$($acc:ident = $i:expr),*]: ntt_i accessors the
    // reference has; [$($m:ident),+]: the mod_p arguments; binary: whether fwd_binary exists; $($derive)*: derives of the reference
    ($(#[$meta:meta])* $name:ident, $kind:expr, $w:ty, $r:ty, $sub:ident, $getter:ident,
     accessors [$($acc:ident = $i:expr),*], residues [$($m:ident),+], binary $binary:tt, rhs $rhs:ident) => {
        $(#[$meta])*
        pub struct $name(NativeRaw);
        impl $name {
            /// `None` where the reference returns `None` (size not a power of two / too small for one of the primes).
            #[track_caller]
            pub fn try_new(n: usize) -> Option<Self> { NativeRaw::try_new($kind, n).map(Self) }
            pub fn ntt_size(&self) -> usize { self.0.ntt_size() }
            $(
                /// Borrowed prime plan of residue `$i` (the reference returns `&Plan`): tied to `&self`, derefs to `Plan`.
                pub fn $acc(&self) -> $sub::PlanRef<'_> { unsafe { $sub::Plan::borrowed(ffi::$getter((self.0).0, $i)) } }
            )*
            /// `residues[i] <- NTT_i(value mod P_i)`; panics unless every slice has `ntt_size()` elements.
            #[track_caller]
            pub fn fwd(&self, value: &[$w], $($m: &mut [$r]),+) {
                $(assert_eq!(value.len(), $m.len());)+
                let r = [$($m.as_mut_ptr() as *mut c_void),+];
                check(unsafe { ffi::cntt_native_fwd((self.0).0, value.as_ptr() as *const c_void, value.len(), r.as_ptr()) })
            }
            native_plan!(@binary $binary, $w, $r, [$($m),+]);
            /// Inverse transforms of the residues IN PLACE (the reference overwrites them too), then CRT into `value`.
            #[track_caller]
            pub fn inv(&self, value: &mut [$w], $($m: &mut [$r]),+) {
                $(assert_eq!(value.len(), $m.len());)+
                let r = [$($m.as_mut_ptr() as *mut c_void),+];
                check(unsafe { ffi::cntt_native_inv((self.0).0, value.as_mut_ptr() as *mut c_void, value.len(), r.as_ptr()) })
            }
            /// Wrapping negacyclic product; panics unless the three lengths are equal (and equal `ntt_size()`).
            #[track_caller]
            pub fn negacyclic_polymul(&self, prod: &mut [$w], lhs: &[$w], $rhs: &[$w]) {
                check(unsafe { ffi::cntt_native_negacyclic_polymul((self.0).0, prod.as_mut_ptr() as *mut c_void, prod.len(),
                                                                   lhs.as_ptr() as *const c_void, lhs.len(), $rhs.as_ptr() as *const c_void, $rhs.len()) })
            }

            // ---- GPU fast path (not in the reference) ----
            /// Grow the plan's device workspace for `batch` products ahead of a timed or captured region.
            pub fn reserve(&self, batch: usize) { self.0.reserve(batch) }
            /// `batch` products, operands back to back in host memory.
            #[track_caller]
            pub fn negacyclic_polymul_batch(&self, prod: &mut [$w], lhs: &[$w], $rhs: &[$w]) {
                assert!(prod.len() == lhs.len() && lhs.len() == $rhs.len() && prod.len() % self.ntt_size() == 0);
                check(unsafe { ffi::cntt_native_negacyclic_polymul_batch((self.0).0, prod.as_mut_ptr() as *mut c_void, lhs.as_ptr() as *const c_void,
                                                                         $rhs.as_ptr() as *const c_void, prod.len() / self.ntt_size(), ffi::CNTT_MEM_HOST, core::ptr::null_mut()) })
            }
            pub unsafe fn negacyclic_polymul_batch_device(&self, prod: *mut $w, lhs: *const $w, $rhs: *const $w, batch: usize, stream: Stream) {
                check(ffi::cntt_native_negacyclic_polymul_batch((self.0).0, prod as *mut c_void, lhs as *const c_void, $rhs as *const c_void, batch, ffi::CNTT_MEM_DEVICE, stream))
            }
            /// Batched `fwd` / `inv` on device memory: `residues[i]` addresses `batch * ntt_size()` residue words.
            pub unsafe fn fwd_batch_device(&self, value: *const $w, residues: &[*mut $r], batch: usize, stream: Stream) {
                assert_eq!(residues.len(), self.0.nprimes());
                let r: Vec<*mut c_void> = residues.iter().map(|p| *p as *mut c_void).collect();
                check(ffi::cntt_native_fwd_batch((self.0).0, value as *const c_void, r.as_ptr(), batch, ffi::CNTT_MEM_DEVICE, stream))
            }
            pub unsafe fn inv_batch_device(&self, value: *mut $w, residues: &[*mut $r], batch: usize, stream: Stream) {
                assert_eq!(residues.len(), self.0.nprimes());
                let r: Vec<*mut c_void> = residues.iter().map(|p| *p as *mut c_void).collect();
                check(ffi::cntt_native_inv_batch((self.0).0, value as *mut c_void, r.as_ptr(), batch, ffi::CNTT_MEM_DEVICE, stream))
            }
            #[doc(hidden)]
            pub fn layout(&self) -> (usize, usize, usize) { (self.0.nprimes(), self.0.word_bytes(), self.0.residue_bytes()) }
        }
    };
    (@binary true, $w:ty, $r:ty, [$($m:ident),+]) => {
        /// `fwd` of a polynomial with coefficients in {0, 1}: the value is copied without `%` (src/native_binary64.rs:379-385).
        #[track_caller]
        pub fn fwd_binary(&self, value: &[$w], $($m: &mut [$r]),+) {
            $(assert_eq!(value.len(), $m.len());)+
            let r = [$($m.as_mut_ptr() as *mut c_void),+];
            check(unsafe { ffi::cntt_native_fwd_binary((self.0).0, value.as_ptr() as *const c_void, value.len(), r.as_ptr()) })
        }
        pub unsafe fn fwd_binary_batch_device(&self, value: *const $w, residues: &[*mut $r], batch: usize, stream: Stream) {
            assert_eq!(residues.len(), self.0.nprimes());
            let r: Vec<*mut c_void> = residues.iter().map(|p| *p as *mut c_void).collect();
            check(ffi::cntt_native_fwd_binary_batch((self.0).0, value as *const c_void, r.as_ptr(), batch, ffi::CNTT_MEM_DEVICE, stream))
        }
    };
    (@binary false, $w:ty, $r:ty, [$($m:ident),+]) => {};
}

// `#[derive(Clone, Debug)]` exactly where the reference derives it (native_binary128::Plan32 derives nothing).
macro_rules! clone_debug {
    ($name:ident) => {
        impl Clone for $name {
            fn clone(&self) -> Self { Self(self.0.clone()) }
        }
        impl core::fmt::Debug for $name {
            fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
                f.debug_struct(stringify!($name)).field("ntt_size", &self.ntt_size()).finish()
            }
        }
    };
}

/// Negacyclic NTT for multiplying two polynomials with values less than `2^32`.
pub mod native32 {
    use super::*;
    native_plan!(/// 3 x 30-bit primes (src/native32.rs:8-12, :335-432)
                 Plan32, 0, u32, u32, prime32, cntt_native_ntt32, accessors [ntt_0 = 0, ntt_1 = 1, ntt_2 = 2],
                 residues [mod_p0, mod_p1, mod_p2], binary false, rhs rhs);
    clone_debug!(Plan32);
    native_plan!(/// 2 x 50-bit primes (src/native32.rs:19, :434-496; the reference offers it only with AVX-512 IFMA)
                 Plan52, 6, u32, u64, prime64, cntt_native_ntt64, accessors [], residues [mod_p0, mod_p1], binary false, rhs rhs);
    clone_debug!(Plan52);
}

/// Negacyclic NTT for multiplying two polynomials with values less than `2^64`.
pub mod native64 {
    use super::*;
    native_plan!(/// 5 x 30-bit primes (src/native64.rs:16-22, :930-1070)
                 Plan32, 1, u64, u32, prime32, cntt_native_ntt32, accessors [ntt_0 = 0, ntt_1 = 1, ntt_2 = 2, ntt_3 = 3, ntt_4 = 4],
                 residues [mod_p0, mod_p1, mod_p2, mod_p3, mod_p4], binary false, rhs rhs);
    clone_debug!(Plan32);
    native_plan!(/// 3 x 50-bit primes (src/native64.rs:29-34, :1074-1165)
                 Plan52, 7, u64, u64, prime64, cntt_native_ntt64, accessors [ntt_0 = 0, ntt_1 = 1, ntt_2 = 2],
                 residues [mod_p0, mod_p1, mod_p2], binary false, rhs rhs);
    clone_debug!(Plan52);
}

/// Negacyclic NTT for multiplying two polynomials with values less than `2^128`.
pub mod native128 {
    use super::*;
    native_plan!(/// 10 x 30-bit primes (src/native128.rs:6-17, :120-349)
                 Plan32, 2, u128, u32, prime32, cntt_native_ntt32,
                 accessors [ntt_0 = 0, ntt_1 = 1, ntt_2 = 2, ntt_3 = 3, ntt_4 = 4, ntt_5 = 5, ntt_6 = 6, ntt_7 = 7, ntt_8 = 8, ntt_9 = 9],
                 residues [mod_p0, mod_p1, mod_p2, mod_p3, mod_p4, mod_p5, mod_p6, mod_p7, mod_p8, mod_p9], binary false, rhs rhs);
    clone_debug!(Plan32);
}

/// Negacyclic NTT for multiplying a polynomial with values less than `2^32` with a binary polynomial.
pub mod native_binary32 {
    use super::*;
    native_plan!(/// 2 x 30-bit primes (src/native_binary32.rs:11, :187-254)
                 Plan32, 3, u32, u32, prime32, cntt_native_ntt32, accessors [], residues [mod_p0, mod_p1], binary true, rhs rhs_binary);
    clone_debug!(Plan32);
    native_plan!(/// 1 x 50-bit prime (src/native_binary32.rs:19, :256-322)
                 Plan52, 8, u32, u64, prime64, cntt_native_ntt64, accessors [], residues [mod_p0], binary true, rhs rhs_binary);
    clone_debug!(Plan52);
}

/// Negacyclic NTT for multiplying a polynomial with values less than `2^64` with a binary polynomial.
pub mod native_binary64 {
    use super::*;
    native_plan!(/// 3 x 30-bit primes (src/native_binary64.rs:17-21, :342-445)
                 Plan32, 4, u64, u32, prime32, cntt_native_ntt32, accessors [], residues [mod_p0, mod_p1, mod_p2], binary true, rhs rhs_binary);
    clone_debug!(Plan32);
    native_plan!(/// 2 x 50-bit primes (src/native_binary64.rs:29, :449-521)
                 Plan52, 9, u64, u64, prime64, cntt_native_ntt64, accessors [], residues [mod_p0, mod_p1], binary true, rhs rhs_binary);
    clone_debug!(Plan52);
}

/// Negacyclic NTT for multiplying a polynomial with values less than `2^128` with a binary polynomial.
pub mod native_binary128 {
    use super::*;
    native_plan!(/// 5 x 30-bit primes (src/native_binary128.rs:4-10, :65-197).  Derives nothing in the reference: no Clone, no Debug.
                 Plan32, 5, u128, u32, prime32, cntt_native_ntt32, accessors [],
                 residues [mod_p0, mod_p1, mod_p2, mod_p3, mod_p4], binary true, rhs rhs);
}

// -------------------------------------------------------------------------------------------------------------------
// product::Plan  (src/product.rs:139-967): negacyclic NTT modulo a product of distinct primes that fits u64
// -------------------------------------------------------------------------------------------------------------------
pub mod product {
    use super::*;
    /// src/product.rs:124-129
    #[derive(Copy, Clone, Debug)]
    pub enum FwdMode { Generic, Bounded(u64) }
    /// src/product.rs:131-136
    #[derive(Copy, Clone, Debug)]
    pub enum InvMode { Replace, Accumulate }
    fn fwd_args(mode: FwdMode) -> (c_int, u64) {
        match mode { FwdMode::Generic => (ffi::CNTT_FWD_GENERIC, 0), FwdMode::Bounded(b) => (ffi::CNTT_FWD_BOUNDED, b) }
    }
    fn inv_arg(mode: InvMode) -> c_int {
        match mode { InvMode::Replace => ffi::CNTT_INV_REPLACE, InvMode::Accumulate => ffi::CNTT_INV_ACCUMULATE }
    }

    pub struct Plan(*mut ffi::cntt_product);
    unsafe impl Send for Plan {}
    unsafe impl Sync for Plan {}
    impl Plan {
        /// `None`: odd size, a zero or repeated factor, product of the factors (1s skipped) != modulus or overflowing
        /// u64, or a factor without a prime plan (src/product.rs:153-247).
        #[track_caller]
        pub fn try_new(polynomial_size: usize, modulus: u64, factors: impl IntoIterator<Item = u64>) -> Option<Self> {
            let f: Vec<u64> = factors.into_iter().collect();
            let mut out = core::ptr::null_mut();
            let rc = unsafe { ffi::cntt_product_plan_new(polynomial_size, modulus, f.as_ptr(), f.len(), &mut out) };
            option_of(rc, out).map(Self)
        }
        pub fn ntt_size(&self) -> usize { unsafe { ffi::cntt_product_ntt_size(self.0) } }
        pub fn modulus(&self) -> u64 { unsafe { ffi::cntt_product_modulus(self.0) } }
        pub fn ntt_domain_len(&self) -> usize { unsafe { ffi::cntt_product_ntt_domain_len(self.0) } }
        // fields that are private in the reference, for parity tests
        pub fn primes(&self) -> Vec<u64> {
            let k = unsafe { ffi::cntt_product_nprimes32(self.0) + ffi::cntt_product_nprimes64(self.0) };
            (0..k).map(|i| unsafe { ffi::cntt_product_prime(self.0, i) }).collect()
        }
        pub fn plan_32(&self) -> Vec<prime32::PlanRef<'_>> {
            (0..unsafe { ffi::cntt_product_nprimes32(self.0) }).map(|i| unsafe { prime32::Plan::borrowed(ffi::cntt_product_ntt32(self.0, i)) }).collect()
        }
        pub fn plan_64(&self) -> Vec<prime64::PlanRef<'_>> {
            (0..unsafe { ffi::cntt_product_nprimes64(self.0) }).map(|i| unsafe { prime64::Plan::borrowed(ffi::cntt_product_ntt64(self.0, i)) }).collect()
        }
        pub fn modular_inverses(&self) -> Vec<u64> {
            let k = self.primes().len();
            let mut v = vec![0u64; k * (k.max(1) - 1) / 2];
            check(unsafe { ffi::cntt_product_modular_inverses(self.0, v.as_mut_ptr(), v.len()) });
            v
        }
        #[track_caller]
        pub fn fwd(&self, ntt: &mut [u64], standard: &[u64], mode: FwdMode) {
            let (m, b) = fwd_args(mode);
            check(unsafe { ffi::cntt_product_fwd(self.0, ntt.as_mut_ptr(), ntt.len(), standard.as_ptr(), standard.len(), m, b) })
        }
        /// Like the reference, leaves the inverse-transformed residues in `ntt`.
        #[track_caller]
        pub fn inv(&self, standard: &mut [u64], ntt: &mut [u64], mode: InvMode) {
            check(unsafe { ffi::cntt_product_inv(self.0, standard.as_mut_ptr(), standard.len(), ntt.as_mut_ptr(), ntt.len(), inv_arg(mode)) })
        }
        #[track_caller]
        pub fn mul_assign_normalize(&self, lhs: &mut [u64], rhs: &[u64]) {
            check(unsafe { ffi::cntt_product_mul_assign_normalize(self.0, lhs.as_mut_ptr(), lhs.len(), rhs.as_ptr(), rhs.len()) })
        }
        #[track_caller]
        pub fn normalize(&self, values: &mut [u64]) { check(unsafe { ffi::cntt_product_normalize(self.0, values.as_mut_ptr(), values.len()) }) }
        #[track_caller]
        pub fn mul_accumulate(&self, acc: &mut [u64], lhs: &[u64], rhs: &[u64]) {
            check(unsafe { ffi::cntt_product_mul_accumulate(self.0, acc.as_mut_ptr(), acc.len(), lhs.as_ptr(), lhs.len(), rhs.as_ptr(), rhs.len()) })
        }

        // ---- batched, device-resident (plane-major NTT domain: include/cntt.h) ----
        pub unsafe fn fwd_batch_device(&self, ntt: *mut u64, standard: *const u64, batch: usize, mode: FwdMode, stream: Stream) {
            let (m, b) = fwd_args(mode);
            check(ffi::cntt_product_fwd_batch(self.0, ntt, standard, batch, m, b, ffi::CNTT_MEM_DEVICE, stream))
        }
        pub unsafe fn inv_batch_device(&self, standard: *mut u64, ntt: *mut u64, batch: usize, mode: InvMode, stream: Stream) {
            check(ffi::cntt_product_inv_batch(self.0, standard, ntt, batch, inv_arg(mode), ffi::CNTT_MEM_DEVICE, stream))
        }
        pub unsafe fn mul_assign_normalize_batch_device(&self, lhs: *mut u64, rhs: *const u64, batch: usize, stream: Stream) {
            check(ffi::cntt_product_mul_assign_normalize_batch(self.0, lhs, rhs, batch, ffi::CNTT_MEM_DEVICE, stream))
        }
        pub unsafe fn normalize_batch_device(&self, values: *mut u64, batch: usize, stream: Stream) {
            check(ffi::cntt_product_normalize_batch(self.0, values, batch, ffi::CNTT_MEM_DEVICE, stream))
        }
        pub unsafe fn mul_accumulate_batch_device(&self, acc: *mut u64, lhs: *const u64, rhs: *const u64, batch: usize, stream: Stream) {
            check(ffi::cntt_product_mul_accumulate_batch(self.0, acc, lhs, rhs, batch, ffi::CNTT_MEM_DEVICE, stream))
        }
        /// The external-product step of tfhe-rs's NTT backend in one call (include/cntt.h).
        pub unsafe fn external_product_batch_device(&self, out: *mut u64, terms: *const u64, key_ntt: *const u64, nterms: usize, nout: usize,
                                                    batch: usize, fwd_mode: FwdMode, inv_mode: InvMode, stream: Stream) {
            let (m, b) = fwd_args(fwd_mode);
            check(ffi::cntt_product_external_product_batch(self.0, out, terms, key_ntt, nterms, nout, batch, m, b, inv_arg(inv_mode), ffi::CNTT_MEM_DEVICE, stream))
        }
    }
    impl Clone for Plan {
        fn clone(&self) -> Self { Self(unsafe { ffi::cntt_product_plan_clone(self.0) }) }
    }
    impl Drop for Plan {
        fn drop(&mut self) { unsafe { ffi::cntt_product_plan_free(self.0) } }
    }
    impl core::fmt::Debug for Plan {
        fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
            f.debug_struct("Plan").field("ntt_size", &self.ntt_size()).field("modulus", &self.modulus()).field("primes", &self.primes()).finish()
        }
    }
}
