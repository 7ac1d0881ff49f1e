//! All GPUs of the node from one process: `bjj_multi_*` (BASELINE configs[4]).  The batch is cut into
//! contiguous ceil(n/G) blocks; host slices go through one pipeline thread per device, device-resident arrays through a
//! pipelined RCCL scatter / kernels / gather inside libbjj_hip.so (peer blocks travel in pieces; `set_chunks`).
use crate::ffi;
use crate::gpu::{check, BorrowedGpu, Gpu};
use std::os::raw::{c_int, c_void};
use std::ptr;

pub struct MultiGpu {
    m: *mut ffi::BjjMulti,
}

pub struct PhaseTimes {
    pub scatter_ms: f64,
    pub compute_ms: f64,
    pub gather_ms: f64,
    pub rccl_version: i32,
    /// start to last event on any device: what the (overlapped) call took; compare with the sum of the three spans
    pub total_ms: f64,
    pub wall_ms: f64,
    /// pieces per peer block actually used
    pub chunks: i32,
}

impl MultiGpu {
    /// `devices`: distinct HIP device indices; empty = every visible device.
    pub fn new(devices: &[i32], window_bits: i32) -> Result<MultiGpu, String> {
        let mut m: *mut ffi::BjjMulti = ptr::null_mut();
        let devs: Vec<c_int> = devices.iter().map(|&d| d as c_int).collect();
        let (p, n) = if devs.is_empty() { (ptr::null(), 0) } else { (devs.as_ptr(), devs.len() as c_int) };
        check(unsafe { ffi::bjj_multi_init(p, n, window_bits as c_int, &mut m) }, "bjj_multi_init")?;
        Ok(MultiGpu { m })
    }

    pub fn size(&self) -> usize {
        unsafe { ffi::bjj_multi_size(self.m) as usize }
    }

    pub fn device(&self, rank: usize) -> i32 {
        unsafe { ffi::bjj_multi_device(self.m, rank as c_int) as i32 }
    }

    /// the per-device context of a rank (owned by the handle: the borrow ends with `&self`)
    pub fn gpu(&self, rank: usize) -> BorrowedGpu<'_> {
        Gpu::borrowed(unsafe { ffi::bjj_multi_ctx(self.m, rank as c_int) })
    }

    /// block [lo, hi) of `rank` in a batch of n items
    pub fn shard_bounds(&self, n: usize, rank: usize) -> (usize, usize) {
        let (mut lo, mut hi) = (0usize, 0usize);
        unsafe { ffi::bjj_shard_bounds(n, self.size() as c_int, rank as c_int, &mut lo, &mut hi) };
        (lo, hi)
    }

    pub fn mul_fixed_base(&self, scalars: &[u8]) -> Result<Vec<u8>, String> {
        let n = scalars.len() / 32;
        let mut out = vec![0u8; n * 64];
        check(unsafe { ffi::bjj_mul_fixed_base_multi(self.m, scalars.as_ptr(), n, out.as_mut_ptr()) }, "bjj_mul_fixed_base_multi")?;
        Ok(out)
    }

    pub fn mul_var_base(&self, points: &[u8], scalars: &[u8]) -> Result<Vec<u8>, String> {
        let n = scalars.len() / 32;
        if points.len() != n * 64 {
            return Err("mul_var_base: array lengths disagree".into());
        }
        let mut out = vec![0u8; n * 64];
        check(
            unsafe { ffi::bjj_mul_var_base_multi(self.m, points.as_ptr(), scalars.as_ptr(), n, out.as_mut_ptr()) },
            "bjj_mul_var_base_multi",
        )?;
        Ok(out)
    }

    pub fn eddsa_verify(&self, pk: &[u8], r_b8: &[u8], s: &[u8], msg: &[u8]) -> Result<Vec<u8>, String> {
        let n = s.len() / 32;
        if pk.len() != n * 64 || r_b8.len() != n * 64 || msg.len() != n * 32 {
            return Err("eddsa_verify: array lengths disagree".into());
        }
        let mut ok = vec![0u8; n];
        check(
            unsafe { ffi::bjj_eddsa_verify_multi(self.m, pk.as_ptr(), r_b8.as_ptr(), s.as_ptr(), msg.as_ptr(), n, ok.as_mut_ptr()) },
            "bjj_eddsa_verify_multi",
        )?;
        Ok(ok)
    }

    /// Device-resident form: the four input arrays and `d_ok` live in the HBM of the handle's first device.
    ///
    /// # Safety
    /// The pointers must be valid device allocations of n * 64 / 64 / 32 / 32 / 1 bytes, 16-byte aligned.
    pub unsafe fn eddsa_verify_dev(&self, d_pk: *const c_void, d_r: *const c_void, d_s: *const c_void, d_msg: *const c_void, n: usize,
                                   d_ok: *mut c_void) -> Result<(), String> {
        check(ffi::bjj_eddsa_verify_multi_dev(self.m, d_pk, d_r, d_s, d_msg, n, d_ok), "bjj_eddsa_verify_multi_dev")
    }

    /// Pipeline depth of the device-resident form: at most `chunks` (1..=16) pieces per peer block, none smaller than
    /// `min_chunk_items`; 1 = the serial scatter -> kernels -> gather schedule.
    pub fn set_chunks(&self, chunks: i32, min_chunk_items: usize) -> Result<(), String> {
        check(unsafe { ffi::bjj_multi_set_chunks(self.m, chunks as c_int, min_chunk_items) }, "bjj_multi_set_chunks")
    }

    pub fn last_timing(&self) -> Result<PhaseTimes, String> {
        let (mut s, mut c, mut g, mut v) = (0f64, 0f64, 0f64, 0 as c_int);
        check(unsafe { ffi::bjj_multi_last_timing(self.m, &mut s, &mut c, &mut g, &mut v) }, "bjj_multi_last_timing")?;
        let (mut tot, mut wall, mut ch) = (0f64, 0f64, 0 as c_int);
        check(unsafe { ffi::bjj_multi_last_overlap(self.m, &mut tot, &mut wall, &mut ch) }, "bjj_multi_last_overlap")?;
        Ok(PhaseTimes { scatter_ms: s, compute_ms: c, gather_ms: g, rccl_version: v as i32, total_ms: tot, wall_ms: wall, chunks: ch as i32 })
    }
}

impl Drop for MultiGpu {
    fn drop(&mut self) {
        if !self.m.is_null() {
            unsafe { ffi::bjj_multi_free(self.m) }
        }
    }
}
