//! One GPU context (`bjj_init` / `bjj_free`) and the batch calls over byte records.
//!
//! Records are what `../include/bjj_hip.h` defines: a field element is the 32-byte little-endian canonical integer
//! (`Fr::into_repr().0` as `[u64; 4]`), a point is `x || y` (64 bytes), a projective point `x || y || z` (96 bytes),
//! a scalar / `s` / `msg` a 32-byte little-endian unsigned integer.
use crate::ffi;
use std::cell::RefCell;
use std::ffi::CStr;
use std::os::raw::c_int;
use std::ptr;
use std::rc::Rc;

pub struct Gpu {
    ctx: *mut ffi::BjjCtx,
    owned: bool,
    /// released pinned buffers, reused by `pinned()` (page-locking memory costs ~60 us per MB: never per call)
    pool: RefCell<Vec<(*mut u8, usize)>>,
}

/// Page-locked host memory from `bjj_host_alloc`: the host-pointer entry points copy such arrays straight over PCIe
/// (no staging copy inside the library).  The `*_batch` functions marshal their `BigInt` / `Fr` records directly into
/// these buffers.  Returned to the context's pool on drop.
pub struct PinnedBuf<'a> {
    gpu: &'a Gpu,
    ptr: *mut u8,
    cap: usize,
    len: usize,
}

impl<'a> std::ops::Deref for PinnedBuf<'a> {
    type Target = [u8];
    fn deref(&self) -> &[u8] {
        unsafe { std::slice::from_raw_parts(self.ptr, self.len) }
    }
}

impl<'a> std::ops::DerefMut for PinnedBuf<'a> {
    fn deref_mut(&mut self) -> &mut [u8] {
        unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}

/// The pool keeps at most this many released buffers / bytes; what does not fit is unpinned at once (a caller whose batch sizes
/// vary must not accumulate one power-of-two buffer per size for the life of the context).
const POOL_MAX_BUFFERS: usize = 16;
const POOL_MAX_BYTES: usize = 1 << 30;

impl<'a> PinnedBuf<'a> {
    /// Zero the buffer now (key material, nonces): pooled buffers are handed out again with their old contents.
    pub fn wipe(&mut self) {
        for b in self.iter_mut() {
            unsafe { std::ptr::write_volatile(b, 0) };
        }
    }
}

impl<'a> Drop for PinnedBuf<'a> {
    fn drop(&mut self) {
        let mut pool = self.gpu.pool.borrow_mut();
        let held: usize = pool.iter().map(|&(_, c)| c).sum();
        if pool.len() < POOL_MAX_BUFFERS && held + self.cap <= POOL_MAX_BYTES {
            pool.push((self.ptr, self.cap));
        } else {
            unsafe { ffi::bjj_host_free(self.gpu.ctx, self.ptr as *mut std::os::raw::c_void) };
        }
    }
}

/// A per-device context that belongs to a `multi::MultiGpu` handle: it cannot outlive the handle (its pool's buffers are
/// released through the context, which `bjj_multi_free` destroys).
pub struct BorrowedGpu<'m> {
    gpu: Gpu,
    _handle: std::marker::PhantomData<&'m ()>,
}

impl<'m> std::ops::Deref for BorrowedGpu<'m> {
    type Target = Gpu;
    fn deref(&self) -> &Gpu {
        &self.gpu
    }
}

pub(crate) fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::bjj_last_error()).to_string_lossy().into_owned() }
}

pub(crate) fn check(rc: c_int, what: &str) -> Result<(), String> {
    if rc == ffi::BJJ_OK {
        Ok(())
    } else {
        Err(format!("{} failed ({}): {}", what, rc, last_error()))
    }
}

impl Gpu {
    /// `window_bits`: 0 = the library default (23-bit windows, 5.9 GB table), `ffi::BJJ_WINDOW_AUTO`, or 4..=28.
    pub fn new(device: i32, window_bits: i32) -> Result<Gpu, String> {
        let mut ctx: *mut ffi::BjjCtx = ptr::null_mut();
        check(unsafe { ffi::bjj_init(device as c_int, window_bits as c_int, &mut ctx) }, "bjj_init")?;
        Ok(Gpu { ctx, owned: true, pool: RefCell::new(Vec::new()) })
    }

    /// A context owned by a `multi::MultiGpu` handle; `'m` is the borrow of that handle.
    pub(crate) fn borrowed<'m>(ctx: *mut ffi::BjjCtx) -> BorrowedGpu<'m> {
        BorrowedGpu { gpu: Gpu { ctx, owned: false, pool: RefCell::new(Vec::new()) }, _handle: std::marker::PhantomData }
    }

    /// `len` bytes of pinned memory (contents unspecified): the smallest released buffer that fits, else a new allocation.
    pub fn pinned(&self, len: usize) -> Result<PinnedBuf<'_>, String> {
        let mut pool = self.pool.borrow_mut();
        let mut best: Option<usize> = None;
        for (i, &(_, cap)) in pool.iter().enumerate() {
            if cap >= len && best.map_or(true, |b| cap < pool[b].1) {
                best = Some(i);
            }
        }
        if let Some(i) = best {
            let (ptr, cap) = pool.swap_remove(i);
            return Ok(PinnedBuf { gpu: self, ptr, cap, len });
        }
        let cap = len.max(4096).next_power_of_two();
        let mut p: *mut std::os::raw::c_void = ptr::null_mut();
        check(unsafe { ffi::bjj_host_alloc(self.ctx, cap, &mut p) }, "bjj_host_alloc")?;
        Ok(PinnedBuf { gpu: self, ptr: p as *mut u8, cap, len })
    }

    pub fn info(&self) -> Result<ffi::BjjInfo, String> {
        let mut i: ffi::BjjInfo = unsafe { std::mem::zeroed() };
        i.struct_size = std::mem::size_of::<ffi::BjjInfo>() as u32;   // the library fills at most this many bytes
        check(unsafe { ffi::bjj_get_info(self.ctx, &mut i) }, "bjj_get_info")?;
        Ok(i)
    }

    /// Signer hardening: `public_keys` / `sign` / `sign_schnorr` then scan a small 4-bit table instead of indexing the big
    /// fixed-base table with digits of the nonce and of the scalar key -- no memory address and no branch depends on a
    /// secret.  Results are bit-identical; `sign` costs about twice as much.  Off by default (the reference branches on
    /// secret bits itself, src/lib.rs:158).  `BJJ_SIGNER_CONSTANT_TIME=1` turns it on for the thread's lazily created context.
    pub fn set_signer_constant_time(&self, on: bool) -> Result<(), String> {
        check(unsafe { ffi::bjj_set_signer_constant_time(self.ctx, on as c_int) }, "bjj_set_signer_constant_time")
    }

    /// `B8.mul_scalar(n)` for every 32-byte scalar (reference src/lib.rs:149-164 with self = B8).
    pub fn mul_fixed_base(&self, scalars: &[u8]) -> Result<Vec<u8>, String> {
        let n = records(scalars, 32, "scalars")?;
        let mut out = vec![0u8; n * 64];
        check(unsafe { ffi::bjj_mul_fixed_base(self.ctx, scalars.as_ptr(), n, out.as_mut_ptr()) }, "bjj_mul_fixed_base")?;
        Ok(out)
    }

    /// The same on caller-provided buffers (pinned ones are copied directly): `out` holds n * 64 bytes.
    pub fn mul_fixed_base_into(&self, scalars: &[u8], out: &mut [u8]) -> Result<(), String> {
        let n = records(scalars, 32, "scalars")?;
        if out.len() != n * 64 {
            return Err("mul_fixed_base_into: out must hold 64 bytes per scalar".into());
        }
        check(unsafe { ffi::bjj_mul_fixed_base(self.ctx, scalars.as_ptr(), n, out.as_mut_ptr()) }, "bjj_mul_fixed_base")
    }

    pub fn mul_var_base_into(&self, points: &[u8], scalars: &[u8], scalar_bytes: usize, out: &mut [u8]) -> Result<(), String> {
        let n = records(points, 64, "points")?;
        if scalar_bytes == 0 || scalar_bytes % 32 != 0 || scalars.len() != n * scalar_bytes || out.len() != n * 64 {
            return Err("mul_var_base_into: array lengths disagree".into());
        }
        let rc = unsafe {
            if scalar_bytes == 32 {
                ffi::bjj_mul_var_base(self.ctx, points.as_ptr(), scalars.as_ptr(), n, out.as_mut_ptr())
            } else {
                ffi::bjj_mul_var_base_wide(self.ctx, points.as_ptr(), scalars.as_ptr(), scalar_bytes, n, out.as_mut_ptr())
            }
        };
        check(rc, "bjj_mul_var_base")
    }

    pub fn eddsa_verify_into(&self, pk: &[u8], r_b8: &[u8], s: &[u8], msg: &[u8], ok: &mut [u8]) -> Result<(), String> {
        let n = records(s, 32, "s")?;
        if pk.len() != n * 64 || r_b8.len() != n * 64 || msg.len() != n * 32 || ok.len() != n {
            return Err("eddsa_verify_into: array lengths disagree".into());
        }
        check(
            unsafe { ffi::bjj_eddsa_verify(self.ctx, pk.as_ptr(), r_b8.as_ptr(), s.as_ptr(), msg.as_ptr(), n, ok.as_mut_ptr()) },
            "bjj_eddsa_verify",
        )
    }

    /// `P.mul_scalar(n)`; `scalar_bytes` per scalar record (a multiple of 32; 32 takes the fast entry point).
    pub fn mul_var_base(&self, points: &[u8], scalars: &[u8], scalar_bytes: usize) -> Result<Vec<u8>, String> {
        let n = records(points, 64, "points")?;
        if scalar_bytes == 0 || scalar_bytes % 32 != 0 || scalars.len() != n * scalar_bytes {
            return Err("mul_var_base: scalar records do not match the points".into());
        }
        let mut out = vec![0u8; n * 64];
        let rc = unsafe {
            if scalar_bytes == 32 {
                ffi::bjj_mul_var_base(self.ctx, points.as_ptr(), scalars.as_ptr(), n, out.as_mut_ptr())
            } else {
                ffi::bjj_mul_var_base_wide(self.ctx, points.as_ptr(), scalars.as_ptr(), scalar_bytes, n, out.as_mut_ptr())
            }
        };
        check(rc, "bjj_mul_var_base")?;
        Ok(out)
    }

    pub fn poseidon5(&self, inputs: &[u8]) -> Result<Vec<u8>, String> {
        let n = records(inputs, 160, "inputs")?;
        let mut out = vec![0u8; n * 32];
        check(unsafe { ffi::bjj_poseidon5(self.ctx, inputs.as_ptr(), n, out.as_mut_ptr()) }, "bjj_poseidon5")?;
        Ok(out)
    }

    /// one verdict byte per signature: 1 = `verify` returned true, 0 = false (reference src/lib.rs:395-412)
    pub fn eddsa_verify(&self, pk: &[u8], r_b8: &[u8], s: &[u8], msg: &[u8]) -> Result<Vec<u8>, String> {
        let n = records(s, 32, "s")?;
        if pk.len() != n * 64 || r_b8.len() != n * 64 || msg.len() != n * 32 {
            return Err("eddsa_verify: array lengths disagree".into());
        }
        let mut ok = vec![0u8; n];
        check(
            unsafe { ffi::bjj_eddsa_verify(self.ctx, pk.as_ptr(), r_b8.as_ptr(), s.as_ptr(), msg.as_ptr(), n, ok.as_mut_ptr()) },
            "bjj_eddsa_verify",
        )?;
        Ok(ok)
    }

    /// 1 / 0 = `Ok(true / false)`, 2 = `Err` (msg > Q); `s` already reduced mod 8l into 32 bytes
    pub fn schnorr_verify(&self, pk: &[u8], r: &[u8], s: &[u8], msg: &[u8]) -> Result<Vec<u8>, String> {
        let n = records(s, 32, "s")?;
        if pk.len() != n * 64 || r.len() != n * 64 || msg.len() != n * 32 {
            return Err("schnorr_verify: array lengths disagree".into());
        }
        let mut ok = vec![0u8; n];
        check(
            unsafe { ffi::bjj_schnorr_verify(self.ctx, pk.as_ptr(), r.as_ptr(), s.as_ptr(), msg.as_ptr(), n, ok.as_mut_ptr()) },
            "bjj_schnorr_verify",
        )?;
        Ok(ok)
    }

    /// raw `PointProjective::add` (reference src/lib.rs:88-131): 96-byte records, any z
    pub fn proj_add(&self, p: &[u8], q: &[u8]) -> Result<Vec<u8>, String> {
        let n = records(p, 96, "p")?;
        if q.len() != p.len() {
            return Err("proj_add: array lengths disagree".into());
        }
        let mut out = vec![0u8; n * 96];
        check(unsafe { ffi::bjj_proj_add(self.ctx, p.as_ptr(), q.as_ptr(), n, out.as_mut_ptr()) }, "bjj_proj_add")?;
        Ok(out)
    }

    /// `PointProjective::affine` (reference src/lib.rs:70-85); z == 0 gives (0, 0)
    pub fn proj_affine(&self, p: &[u8]) -> Result<Vec<u8>, String> {
        let n = records(p, 96, "p")?;
        let mut out = vec![0u8; n * 64];
        check(unsafe { ffi::bjj_proj_affine(self.ctx, p.as_ptr(), n, out.as_mut_ptr()) }, "bjj_proj_affine")?;
        Ok(out)
    }

    pub fn compress_points(&self, pts: &[u8]) -> Result<Vec<u8>, String> {
        let n = records(pts, 64, "points")?;
        let mut out = vec![0u8; n * 32];
        check(unsafe { ffi::bjj_compress_points(self.ctx, pts.as_ptr(), n, out.as_mut_ptr()) }, "bjj_compress_points")?;
        Ok(out)
    }

    /// (points, ok): ok[i] == 0 where `decompress_point` returns `Err` (reference src/lib.rs:192-224)
    pub fn decompress_points(&self, comp: &[u8]) -> Result<(Vec<u8>, Vec<u8>), String> {
        let n = records(comp, 32, "compressed points")?;
        let (mut out, mut ok) = (vec![0u8; n * 64], vec![0u8; n]);
        check(
            unsafe { ffi::bjj_decompress_points(self.ctx, comp.as_ptr(), n, out.as_mut_ptr(), ok.as_mut_ptr()) },
            "bjj_decompress_points",
        )?;
        Ok((out, ok))
    }

    pub fn scalar_keys(&self, keys: &[u8]) -> Result<Vec<u8>, String> {
        let n = records(keys, 32, "keys")?;
        let mut out = vec![0u8; n * 32];
        check(unsafe { ffi::bjj_scalar_keys(self.ctx, keys.as_ptr(), n, out.as_mut_ptr()) }, "bjj_scalar_keys")?;
        Ok(out)
    }

    pub fn public_keys(&self, keys: &[u8]) -> Result<Vec<u8>, String> {
        let n = records(keys, 32, "keys")?;
        let mut out = vec![0u8; n * 64];
        check(unsafe { ffi::bjj_public_keys(self.ctx, keys.as_ptr(), n, out.as_mut_ptr()) }, "bjj_public_keys")?;
        Ok(out)
    }

    /// `sk.public().compress()` for every key in one pass (reference src/lib.rs:304-306 + 166-178): 32 bytes per key, the
    /// compression fused into the multiplication's epilogue -- half the bytes back across PCIe and no second call.
    pub fn public_keys_compressed(&self, keys: &[u8]) -> Result<Vec<u8>, String> {
        let n = records(keys, 32, "keys")?;
        let mut out = vec![0u8; n * 32];
        check(unsafe { ffi::bjj_public_keys_compressed(self.ctx, keys.as_ptr(), n, out.as_mut_ptr()) }, "bjj_public_keys_compressed")?;
        Ok(out)
    }

    /// `B8.mul_scalar(n).compress()` for every 32-byte scalar in one pass (reference src/lib.rs:149-164 + 166-178)
    pub fn mul_fixed_base_compressed(&self, scalars: &[u8]) -> Result<Vec<u8>, String> {
        let n = records(scalars, 32, "scalars")?;
        let mut out = vec![0u8; n * 32];
        check(unsafe { ffi::bjj_mul_fixed_base_compressed(self.ctx, scalars.as_ptr(), n, out.as_mut_ptr()) }, "bjj_mul_fixed_base_compressed")?;
        Ok(out)
    }

    /// (sig, ok): `sk.sign(msg)?.compress()` (reference src/lib.rs:308-342 + 245-258), 64 bytes per signature; ok[i] == 0 (and an
    /// all-zero record) where `PrivateKey::sign` returns `Err`
    pub fn sign_compressed(&self, keys: &[u8], msgs: &[u8]) -> Result<(Vec<u8>, Vec<u8>), String> {
        let n = records(keys, 32, "keys")?;
        if msgs.len() != n * 32 {
            return Err("sign_compressed: array lengths disagree".into());
        }
        let (mut sig, mut ok) = (vec![0u8; n * 64], vec![0u8; n]);
        check(
            unsafe { ffi::bjj_sign_compressed(self.ctx, keys.as_ptr(), msgs.as_ptr(), n, sig.as_mut_ptr(), ok.as_mut_ptr()) },
            "bjj_sign_compressed",
        )?;
        Ok((sig, ok))
    }

    /// (R, s, ok): ok[i] == 0 where `PrivateKey::sign` returns `Err` (msg > Q, reference src/lib.rs:309-311)
    pub fn sign(&self, keys: &[u8], msgs: &[u8]) -> Result<(Vec<u8>, Vec<u8>, Vec<u8>), String> {
        let n = records(keys, 32, "keys")?;
        if msgs.len() != n * 32 {
            return Err("sign: array lengths disagree".into());
        }
        let (mut r, mut s, mut ok) = (vec![0u8; n * 64], vec![0u8; n * 32], vec![0u8; n]);
        check(
            unsafe { ffi::bjj_sign(self.ctx, keys.as_ptr(), msgs.as_ptr(), n, r.as_mut_ptr(), s.as_mut_ptr(), ok.as_mut_ptr()) },
            "bjj_sign",
        )?;
        Ok((r, s, ok))
    }

    /// `PrivateKey::sign_schnorr` with caller-supplied 1024-bit nonces (128 bytes each); s = 160-byte unreduced integers
    pub fn sign_schnorr(&self, keys: &[u8], msgs: &[u8], nonces: &[u8]) -> Result<(Vec<u8>, Vec<u8>, Vec<u8>), String> {
        let n = records(keys, 32, "keys")?;
        if msgs.len() != n * 32 || nonces.len() != n * ffi::BJJ_SCHNORR_NONCE_BYTES {
            return Err("sign_schnorr: array lengths disagree".into());
        }
        let (mut r, mut s, mut ok) = (vec![0u8; n * 64], vec![0u8; n * ffi::BJJ_SCHNORR_S_BYTES], vec![0u8; n]);
        check(
            unsafe {
                ffi::bjj_sign_schnorr(self.ctx, keys.as_ptr(), msgs.as_ptr(), nonces.as_ptr(), n, r.as_mut_ptr(), s.as_mut_ptr(), ok.as_mut_ptr())
            },
            "bjj_sign_schnorr",
        )?;
        Ok((r, s, ok))
    }
}

impl Drop for Gpu {
    fn drop(&mut self) {
        for (p, _) in self.pool.borrow_mut().drain(..) {
            unsafe { ffi::bjj_host_free(self.ctx, p as *mut std::os::raw::c_void) };
        }
        if self.owned && !self.ctx.is_null() {
            unsafe { ffi::bjj_free(self.ctx) }
        }
    }
}

fn records(bytes: &[u8], width: usize, name: &str) -> Result<usize, String> {
    if bytes.len() % width != 0 {
        return Err(format!("{}: byte length {} is not a multiple of {}", name, bytes.len(), width));
    }
    Ok(bytes.len() / width)
}

thread_local! {
    /// The context behind the reference-shaped single-item API: created on first use, one per thread (a `bjj_ctx` is
    /// used by one host thread at a time).  BJJ_DEVICE / BJJ_WINDOW_BITS select the device and the table width,
    /// BJJ_SIGNER_CONSTANT_TIME=1 the hardened signer.
    static GPU: RefCell<Option<Rc<Gpu>>> = RefCell::new(None);
}

fn env_i32(name: &str, default: i32) -> i32 {
    std::env::var(name).ok().and_then(|v| v.parse().ok()).unwrap_or(default)
}

/// Runs `f` with this thread's GPU context.  Panics if no GPU is usable: like the reference, the path functions are
/// infallible, and there is no CPU fallback to fall back to.
pub fn with_gpu<R>(f: impl FnOnce(&Gpu) -> R) -> R {
    let gpu = GPU.with(|cell| {
        let mut slot = cell.borrow_mut();
        if slot.is_none() {
            let g = Gpu::new(env_i32("BJJ_DEVICE", 0), env_i32("BJJ_WINDOW_BITS", 0))
                .unwrap_or_else(|e| panic!("babyjubjub_rs (HIP): {}", e));
            if env_i32("BJJ_SIGNER_CONSTANT_TIME", 0) != 0 {
                g.set_signer_constant_time(true).unwrap_or_else(|e| panic!("babyjubjub_rs (HIP): {}", e));
            }
            *slot = Some(Rc::new(g));
        }
        slot.as_ref().unwrap().clone()
    });
    f(&gpu)
}
