//! `extern "C"` declarations of libbjj_hip.so -- GENERATED from ../include/bjj_hip.h by tools/gen_rust_ffi.py.
//! Do not edit by hand; `tests/test_rust_shim.py` compares this block with the header.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

/// opaque `bjj_ctx` (one device + stream + fixed-base table)
#[repr(C)]
pub struct BjjCtx {
    _private: [u8; 0],
}
/// opaque `bjj_multi` (one context per device of the node)
#[repr(C)]
pub struct BjjMulti {
    _private: [u8; 0],
}

/// `bjj_info` (include/bjj_hip.h): set `struct_size = size_of::<BjjInfo>()` before `bjj_get_info`
#[repr(C)]
pub struct BjjInfo {
    pub struct_size: u32,
    pub device: c_int,
    pub compute_units: c_int,
    pub window_bits: c_int,
    pub n_windows: c_int,
    pub table_bytes: u64,
    pub scratch_bytes: u64,
    pub kernel_fixed_base: *const c_char,
    pub kernel_var_base: *const c_char,
    pub kernel_poseidon5: *const c_char,
    pub kernel_verify: *const c_char,
    pub init_ms: f64,
    pub signer_constant_time: c_int,
    pub last_fixed_base_shape: c_int,
    pub last_var_base_form: c_int,
    pub last_verify_dispatch: c_int,
    pub last_host_direct_arrays: u32,
    pub last_host_staged_arrays: u32,
    pub last_host_chunks: u32,
    pub host_copy_threads: c_int,
    pub kernel_fixed_base_overlap: *const c_char,
    pub kernel_var_base_overlap: *const c_char,
    pub last_var_base_split: c_int,
    pub last_host_zero_copy: u32,
    pub last_poseidon_form: c_int,
    pub last_sign_form: c_int,
}

pub const BJJ_OK: c_int = 0;
pub const BJJ_E_INVALID: c_int = -1;
pub const BJJ_E_NO_DEVICE: c_int = -2;
pub const BJJ_E_HIP: c_int = -3;
pub const BJJ_E_NOMEM: c_int = -4;
pub const BJJ_E_RCCL: c_int = -5;
pub const BJJ_WINDOW_AUTO: c_int = -1;
pub const BJJ_MAX_SCALAR_BYTES: usize = 4096;
pub const BJJ_SCHNORR_NONCE_BYTES: usize = 128;
pub const BJJ_SCHNORR_S_BYTES: usize = 160;
pub const BJJ_TRANSPORT_RCCL: c_int = 0;
pub const BJJ_TRANSPORT_PEER_COPY: c_int = 1;

extern "C" {
    pub fn bjj_version() -> *const c_char;
    pub fn bjj_last_error() -> *const c_char;
    pub fn bjj_init(device: c_int, window_bits: c_int, out_ctx: *mut *mut BjjCtx) -> c_int;
    pub fn bjj_free(ctx: *mut BjjCtx);
    pub fn bjj_sync(ctx: *mut BjjCtx) -> c_int;
    pub fn bjj_stream(ctx: *mut BjjCtx) -> *mut c_void;
    pub fn bjj_host_alloc(ctx: *mut BjjCtx, bytes: usize, out_ptr: *mut *mut c_void) -> c_int;
    pub fn bjj_host_free(ctx: *mut BjjCtx, ptr: *mut c_void) -> c_int;
    pub fn bjj_host_register(ctx: *mut BjjCtx, ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn bjj_host_unregister(ctx: *mut BjjCtx, ptr: *mut c_void) -> c_int;
    pub fn bjj_host_is_pinned(ctx: *mut BjjCtx, ptr: *const c_void, bytes: usize) -> c_int;
    pub fn bjj_mul_fixed_base(ctx: *mut BjjCtx, scalars: *const u8, n: usize, out_xy: *mut u8) -> c_int;
    pub fn bjj_mul_var_base(ctx: *mut BjjCtx, pts_xy: *const u8, scalars: *const u8, n: usize, out_xy: *mut u8) -> c_int;
    pub fn bjj_mul_var_base_wide(ctx: *mut BjjCtx, pts_xy: *const u8, scalars: *const u8, scalar_bytes: usize, n: usize, out_xy: *mut u8) -> c_int;
    pub fn bjj_poseidon5(ctx: *mut BjjCtx, input: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn bjj_eddsa_verify(ctx: *mut BjjCtx, pk_xy: *const u8, r_xy: *const u8, s: *const u8, msg: *const u8, n: usize, ok: *mut u8) -> c_int;
    pub fn bjj_schnorr_verify(ctx: *mut BjjCtx, pk_xy: *const u8, r_xy: *const u8, s: *const u8, msg: *const u8, n: usize, ok: *mut u8) -> c_int;
    pub fn bjj_point_add(ctx: *mut BjjCtx, p_xy: *const u8, q_xy: *const u8, n: usize, out_xy: *mut u8) -> c_int;
    pub fn bjj_proj_add(ctx: *mut BjjCtx, p_xyz: *const u8, q_xyz: *const u8, n: usize, out_xyz: *mut u8) -> c_int;
    pub fn bjj_proj_affine(ctx: *mut BjjCtx, p_xyz: *const u8, n: usize, out_xy: *mut u8) -> c_int;
    pub fn bjj_compress_points(ctx: *mut BjjCtx, pts_xy: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn bjj_mul_fixed_base_compressed(ctx: *mut BjjCtx, scalars: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn bjj_decompress_points(ctx: *mut BjjCtx, input: *const u8, n: usize, out_xy: *mut u8, ok: *mut u8) -> c_int;
    pub fn bjj_eddsa_verify_compressed(ctx: *mut BjjCtx, pk: *const u8, sig: *const u8, msg: *const u8, n: usize, ok: *mut u8) -> c_int;
    pub fn bjj_scalar_keys(ctx: *mut BjjCtx, keys: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn bjj_set_signer_constant_time(ctx: *mut BjjCtx, on: c_int) -> c_int;
    pub fn bjj_public_keys(ctx: *mut BjjCtx, keys: *const u8, n: usize, out_xy: *mut u8) -> c_int;
    pub fn bjj_public_keys_compressed(ctx: *mut BjjCtx, keys: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn bjj_sign(ctx: *mut BjjCtx, keys: *const u8, msgs: *const u8, n: usize, out_r_xy: *mut u8, out_s: *mut u8, ok: *mut u8) -> c_int;
    pub fn bjj_sign_compressed(ctx: *mut BjjCtx, keys: *const u8, msgs: *const u8, n: usize, out_sig: *mut u8, ok: *mut u8) -> c_int;
    pub fn bjj_sign_schnorr(ctx: *mut BjjCtx, keys: *const u8, msgs: *const u8, nonces: *const u8, n: usize, out_r_xy: *mut u8, out_s: *mut u8, ok: *mut u8) -> c_int;
    pub fn bjj_mul_fixed_base_dev(ctx: *mut BjjCtx, d_scalars: *const c_void, n: usize, d_out_xy: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_mul_var_base_dev(ctx: *mut BjjCtx, d_pts_xy: *const c_void, d_scalars: *const c_void, n: usize, d_out_xy: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_mul_var_base_wide_dev(ctx: *mut BjjCtx, d_pts_xy: *const c_void, d_scalars: *const c_void, scalar_bytes: usize, n: usize, d_out_xy: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_poseidon5_dev(ctx: *mut BjjCtx, d_in: *const c_void, n: usize, d_out: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_eddsa_verify_dev(ctx: *mut BjjCtx, d_pk_xy: *const c_void, d_r_xy: *const c_void, d_s: *const c_void, d_msg: *const c_void, n: usize, d_ok: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_schnorr_verify_dev(ctx: *mut BjjCtx, d_pk_xy: *const c_void, d_r_xy: *const c_void, d_s: *const c_void, d_msg: *const c_void, n: usize, d_ok: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_point_add_dev(ctx: *mut BjjCtx, d_p_xy: *const c_void, d_q_xy: *const c_void, n: usize, d_out_xy: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_proj_add_dev(ctx: *mut BjjCtx, d_p_xyz: *const c_void, d_q_xyz: *const c_void, n: usize, d_out_xyz: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_proj_affine_dev(ctx: *mut BjjCtx, d_p_xyz: *const c_void, n: usize, d_out_xy: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_scalar_keys_dev(ctx: *mut BjjCtx, d_keys: *const c_void, n: usize, d_out: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_public_keys_dev(ctx: *mut BjjCtx, d_keys: *const c_void, n: usize, d_out_xy: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_sign_dev(ctx: *mut BjjCtx, d_keys: *const c_void, d_msgs: *const c_void, n: usize, d_out_r_xy: *mut c_void, d_out_s: *mut c_void, d_ok: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_sign_schnorr_dev(ctx: *mut BjjCtx, d_keys: *const c_void, d_msgs: *const c_void, d_nonces: *const c_void, n: usize, d_out_r_xy: *mut c_void, d_out_s: *mut c_void, d_ok: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_compress_points_dev(ctx: *mut BjjCtx, d_pts_xy: *const c_void, n: usize, d_out: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_decompress_points_dev(ctx: *mut BjjCtx, d_in: *const c_void, n: usize, d_out_xy: *mut c_void, d_ok: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_eddsa_verify_compressed_dev(ctx: *mut BjjCtx, d_pk: *const c_void, d_sig: *const c_void, d_msg: *const c_void, n: usize, d_ok: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_mul_fixed_base_compressed_dev(ctx: *mut BjjCtx, d_scalars: *const c_void, n: usize, d_out: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_public_keys_compressed_dev(ctx: *mut BjjCtx, d_keys: *const c_void, n: usize, d_out: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_sign_compressed_dev(ctx: *mut BjjCtx, d_keys: *const c_void, d_msgs: *const c_void, n: usize, d_out_sig: *mut c_void, d_ok: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bjj_reserve(ctx: *mut BjjCtx, n: usize) -> c_int;
    pub fn bjj_check_table(ctx: *mut BjjCtx, n_bad: *mut u64) -> c_int;
    pub fn bjj_get_info(ctx: *mut BjjCtx, info: *mut BjjInfo) -> c_int;
    pub fn bjj_multi_init(devices: *const c_int, n_devices: c_int, window_bits: c_int, out: *mut *mut BjjMulti) -> c_int;
    pub fn bjj_multi_set_transport(m: *mut BjjMulti, transport: c_int) -> c_int;
    pub fn bjj_multi_set_chunks(m: *mut BjjMulti, chunks: c_int, min_chunk_items: usize) -> c_int;
    pub fn bjj_multi_free(m: *mut BjjMulti);
    pub fn bjj_multi_size(m: *const BjjMulti) -> c_int;
    pub fn bjj_multi_ctx(m: *mut BjjMulti, rank: c_int) -> *mut BjjCtx;
    pub fn bjj_multi_device(m: *const BjjMulti, rank: c_int) -> c_int;
    pub fn bjj_shard_bounds(n: usize, n_devices: c_int, rank: c_int, lo: *mut usize, hi: *mut usize);
    pub fn bjj_mul_fixed_base_multi(m: *mut BjjMulti, scalars: *const u8, n: usize, out_xy: *mut u8) -> c_int;
    pub fn bjj_mul_var_base_multi(m: *mut BjjMulti, pts_xy: *const u8, scalars: *const u8, n: usize, out_xy: *mut u8) -> c_int;
    pub fn bjj_eddsa_verify_multi(m: *mut BjjMulti, pk_xy: *const u8, r_xy: *const u8, s: *const u8, msg: *const u8, n: usize, ok: *mut u8) -> c_int;
    pub fn bjj_mul_fixed_base_multi_dev(m: *mut BjjMulti, d_scalars: *const c_void, n: usize, d_out_xy: *mut c_void) -> c_int;
    pub fn bjj_mul_var_base_multi_dev(m: *mut BjjMulti, d_pts_xy: *const c_void, d_scalars: *const c_void, n: usize, d_out_xy: *mut c_void) -> c_int;
    pub fn bjj_eddsa_verify_multi_dev(m: *mut BjjMulti, d_pk_xy: *const c_void, d_r_xy: *const c_void, d_s: *const c_void, d_msg: *const c_void, n: usize, d_ok: *mut c_void) -> c_int;
    pub fn bjj_multi_last_timing(m: *mut BjjMulti, scatter_ms: *mut f64, compute_ms: *mut f64, gather_ms: *mut f64, rccl_version: *mut c_int) -> c_int;
    pub fn bjj_multi_last_overlap(m: *mut BjjMulti, total_ms: *mut f64, wall_ms: *mut f64, chunks: *mut c_int) -> c_int;
}
