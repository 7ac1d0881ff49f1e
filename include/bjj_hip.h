/*
 * bjj_hip.h -- C ABI of libbjj_hip.so: batched BabyJubJub scalar multiplication,
 * Poseidon(t=6) and EdDSA-Poseidon verification on AMD MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE path of the Rust crate
 * arnaucube/babyjubjub-rs v0.0.11 (reference checkout: /root/reference).  The
 * crate has no FFI of its own; its public Rust API is the interface, and each
 * entry point below is the batch form of the function(s) it replaces:
 *
 *   bjj_mul_fixed_base   B8.mul_scalar(n)            src/lib.rs:149-164 with self = B8
 *                        (src/lib.rs:37-46); the engine of PrivateKey::public(),
 *                        src/lib.rs:304-306
 *   bjj_mul_var_base     Point::mul_scalar(&self, n) src/lib.rs:149-164
 *   bjj_poseidon5        POSEIDON.hash(vec![a,b,c,d,e]) as called at
 *                        src/lib.rs:400-404 (poseidon-rs 0.0.8, Cargo.toml:20)
 *   bjj_eddsa_verify     verify(pk, sig, msg)        src/lib.rs:395-412
 *   bjj_schnorr_verify   verify_schnorr(pk, m, r, s) src/lib.rs:375-385 (+ schnorr_hash :364-373)
 *   bjj_mul_var_base_wide  the same for scalars wider than 256 bits (`n: &BigInt` is unbounded,
 *                        src/lib.rs:149, 156-157): records of scalar_bytes = 32 k bytes
 *   bjj_point_add        PointProjective::add(..).affine()
 *                        src/lib.rs:88-131 + 70-85 on affine inputs (z = 1)
 *   bjj_proj_add         PointProjective::add(&self, q)  src/lib.rs:88-131: raw (x, y, z) in and out, any z
 *                        (what the reference's own test chains, src/lib.rs:513-516)
 *   bjj_proj_affine      PointProjective::affine(&self)  src/lib.rs:70-85  (z == 0 -> (0, 0))
 *   bjj_compress_points  Point::compress(&self)      src/lib.rs:166-178
 *   bjj_decompress_points decompress_point(bb)       src/lib.rs:192-224 (+ utils.rs modinv/modsqrt)
 *   bjj_scalar_keys      PrivateKey::scalar_key()    src/lib.rs:284-302 (Blake-512, prune, >> 3)
 *   bjj_public_keys      PrivateKey::public()        src/lib.rs:304-306
 *   bjj_sign             PrivateKey::sign(msg)       src/lib.rs:308-342
 *   bjj_sign_schnorr     PrivateKey::sign_schnorr(m) src/lib.rs:344-361 (caller-supplied nonce)
 *   bjj_eddsa_verify_compressed  decompress_point(pk), decompress_signature(sig)
 *                        (src/lib.rs:260-268), then verify -- the wire-format ingest path
 *   bjj_mul_fixed_base_compressed  B8.mul_scalar(n).compress()         src/lib.rs:149-164 + 166-178
 *   bjj_public_keys_compressed     PrivateKey::public().compress()     src/lib.rs:304-306 + 166-178
 *   bjj_sign_compressed            PrivateKey::sign(msg)?.compress()   src/lib.rs:308-342 + 245-258
 *                        -- the wire-format OUTPUT path: the compression is fused into the producing kernel's
 *                        epilogue, results are 32 (64) bytes instead of 64 (96) on their way across PCIe
 *
 * Data formats (all little-endian, caller-owned, tightly packed arrays):
 *   field element  32 bytes, the canonical integer < r  (== Fr::into_repr().0
 *                  as [u64;4]); values >= r on input are reduced mod r
 *   point          64 bytes: x then y  (the crate's `Point { x, y }`,
 *                  src/lib.rs:134-138).  Points are NOT required to be on the
 *                  curve: like the reference, off-curve inputs are processed with
 *                  the reference's exact formula sequence
 *   scalar / s / msg  32 bytes, unsigned integer (BigInt::to_bytes_le zero-padded
 *                  to 32 bytes, as src/lib.rs:249-252 does for `s`)
 *   projective point  96 bytes: x, y, z  (the crate's `PointProjective { x, y, z }`, src/lib.rs:62-67)
 *   ok             1 byte per item: 1 = verify() returned true, 0 = false
 *
 * Domain restrictions of the boundary (everything else is total, like the reference):
 *   bjj_point_add takes affine operands (z = 1) and returns the affine sum -- use bjj_proj_add /
 *   bjj_proj_affine for general z and the raw (x, y, z) result;  32-byte scalar entry points
 *   take n < 2^256 -- use bjj_mul_var_base_wide beyond;  negative integers do not exist at this
 *   boundary (the reference drops the sign of n, src/lib.rs:156, and panics on a negative msg, :399).
 *
 * Error model: the reference's functions on this path are infallible
 * (mul_scalar, add) or fold every failure into `false` (verify, src/lib.rs:396-404).
 * Accordingly per-item outcomes are DATA (ok[i]); the int status is only for
 * API misuse / HIP runtime errors.  0 = success; negative = BJJ_E_*; see
 * bjj_last_error().
 *
 * Batch size: any n < 2^32 per call (byte offsets are 64-bit; item indices travel as 32-bit words).
 *
 * Threading: a context is bound to one device and one internal stream; calls on
 * the same context are serialised by the caller (one host thread at a time);
 * different contexts are independent, also across devices of one process: every
 * entry point runs on its context's device and restores the calling thread's
 * current HIP device before it returns.  *_dev calls return before the work has
 * run.  A context keeps TWO sets of scratch: a stream keeps the set it used
 * last, a second stream gets the other one, and launches on the two streams run
 * concurrently (alternating two streams is worth +4..7 % throughput for every
 * kernel of 2^20-item batches -- the library then also switches the kernels to
 * the forms that overlap best).  A third stream takes
 * over the least recently used set and first waits, on the device, for that
 * set's last call.  The library keeps no pointer past return.
 * Environment knobs (read by bjj_init; for tests and A/B runs): BJJ_K1_VARIANT,
 * BJJ_K2_VARIANT, BJJ_VERIFY_DISPATCH = 0 | 1 force one form of the fixed-base /
 * variable-base / verify kernel instead of the per-call choice; BJJ_VB_SPLIT = 0 | 1
 * forces where the exact kernel of the variable-base path runs (below).
 *
 * Malformed points on the variable-base path.  Point has pub fields and no check (src/lib.rs:134-138), so an
 * (x, y) that is not on the curve is a legal input; its result is whatever the reference's formula sequence
 * yields, and the library reproduces it bit for bit with a strictly serial replay of that loop (~2.1 ms for one
 * item on a pair of lanes, whatever the batch size).  bjj_mul_var_base(_wide)_dev runs that exact kernel BEHIND the
 * batch kernel while every completed call of the context has been clean (a clean batch pays nothing), and -- from
 * the first call after one that met an off-curve point -- BESIDE it on a priority stream, behind a scan of the
 * points (0.7 % of a launch): the malformed items then cost their share of the chip instead of a launch-long tail
 * (2^20 items, 1 off-curve point in 4 096: see profiles/r06_var_base_offcurve.txt).  The host-pointer forms of more
 * than one pipeline chunk always run the exact items as ONE launch beside the chunks' launches and lay their
 * results over the caller's array at the end.  bjj_get_info: last_var_base_split.
 *
 * Key material: the signer-side entry points wipe every library-owned buffer the
 * keys / nonces passed through (staging buffers, derived scalar keys) before they
 * return, and bjj_free zeroes what is left.  They are NOT constant-time: the
 * fixed-base table is indexed with digits of the secret scalar.
 *
 * Host entry points take host pointers and do H2D / kernel / D2H synchronously, as a chunked pipeline (copy in, kernels and
 * copy out of consecutive chunks overlap).  Arrays in PINNED host memory -- bjj_host_alloc, bjj_host_register, or any range the
 * HIP runtime knows as pinned (hipHostMalloc, hipHostRegister, torch's pin_memory) -- are copied from / to directly; pageable
 * arrays are staged through pinned buffers by worker threads of the context (BJJ_STAGE_THREADS, default 4).  The choice is per
 * array and per call; results are identical.  2^20 fixed-base multiplications: about 1.6 ms pinned (the 64 MB of results
 * crossing PCIe take 1.19 ms), 2.0 to 2.6 ms pageable, 0.6 ms on device pointers.
 * The fixed-base and key entry points skip the copy-in for short calls on pinned, 16-byte aligned inputs (one or two pipeline
 * chunks: up to ~2^17 items): the first kernel reads the caller's array through its device mapping, ~15 us less per call
 * (bjj_info.last_host_zero_copy, bit 1).  The array must not be written while the call runs -- as for every host-pointer call.
 * *_dev entry points take DEVICE pointers (16-byte aligned) plus a hipStream_t
 * (passed as void*; NULL = the context's stream), enqueue the work and return
 * without synchronising -- they are what bench.py times.
 * The verifiers on host pointers run the items whose pk or R is off the curve (the reference's exact formula sequence, ~3x as
 * long) as ONE launch for the whole call beside the chunks' launches, and their verdicts leave the device once, at the end:
 * 2^20 signatures of which 1 in 128 is off the curve take about 19 ms pinned, 18.2 ms as one device-pointer launch.
 * Short calls -- at most 2^15 fixed-base multiplications or public keys, 2^14 variable-base multiplications (32-byte scalars) or Poseidon hashes, 2^13 EdDSA
 * verifications -- run kernels that spread one item over four / six / eight lanes (csrc/k_small.hip): identical results, 0.10 / 0.50 / 0.22 / 0.59 ms per call instead of
 * 0.13 / 1.2 / 0.55 / 1.45
 * (a single Point::mul_scalar, POSEIDON.hash or verify of the reference is such a call); bjj_sign / bjj_sign_compressed of at most 2^13 signatures likewise
 * (0.38 ms per call instead of 0.7).  BJJ_FB_QUAD_MAX, BJJ_VB_QUAD_MAX, BJJ_P5_COOP_MAX, BJJ_VERIFY_SMALL_MAX, BJJ_SIGN_SMALL_MAX (items, read at bjj_init; 0 = never) move the
 * switch-overs.
 * More environment knobs (read when the context first runs a host-pointer call): BJJ_PIPE_CHUNK / BJJ_PIPE_FIRST_CHUNK
 * (items per pipeline chunk: the first chunk, doubling up to the cap; defaults 32768 / 131072 -- 65536 / 131072 (262144 beyond 1.5 M items) for the compressed fixed-base and key forms, whose chunk launches take one workgroup slot per CU each --, for the verifiers 65536 /
 * 524288 and for the variable-base multiplications 65536 / 262144; a value in the environment applies to all), BJJ_PIPE_STAGING_MB
 * (device staging a call may take, default 1024; larger batches run as consecutive super-batches), BJJ_HOST_FORCE_STAGED=1
 * (treat every host array as pageable), BJJ_STAGE_THREADS.
 */
#ifndef BJJ_HIP_H
#define BJJ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bjj_ctx bjj_ctx;

enum {
  BJJ_OK = 0,
  BJJ_E_INVALID = -1,   /* bad argument (NULL pointer, misaligned device pointer, ...) */
  BJJ_E_NO_DEVICE = -2, /* no usable HIP device: there is NO CPU fallback */
  BJJ_E_HIP = -3,       /* HIP runtime error, text in bjj_last_error() */
  BJJ_E_NOMEM = -4,
  BJJ_E_RCCL = -5       /* RCCL could not be loaded / a collective failed (multi-GPU entry points) */
};

#define BJJ_WINDOW_AUTO (-1)        /* bjj_init: widest fixed-base table that fits in 60 % of the free HBM */
#define BJJ_MAX_SCALAR_BYTES 4096   /* bjj_mul_var_base_wide: scalars up to 32768 bits */
#define BJJ_MAX_DEVICES 64

/* Library / build identification: "bjj-hip <version> gfx950". */
const char* bjj_version(void);
/* Text of the last error on this thread ("" if none). */
const char* bjj_last_error(void);

/* Creates a context on HIP device `device`: stream, scratch, and the fixed-base
 * window table of B8 multiples (built on the GPU).  `window_bits` selects the
 * fixed-base window width W: scalars are reduced mod l and recoded into
 * ceil(252/W) signed digits, the table holds (2^(W-1) + 1) entries of 128 bytes per
 * window, resident in HBM (W = 16: 67 MB, 21: 1.6 GB, 23: 5.9 GB, 26: 43 GB, 28: 155 GB).
 * 0 = the default, W = 23 (5.9 GB): a modest footprint for a library that shares the GPU with a
 * prover.  The wide tables are opt-in: pass 26 / 28 explicitly (28 is what bench.py measures: one
 * addition less per digit saved), or BJJ_WINDOW_AUTO = the widest of 28 / 26 / 23 / 21 / 16 whose
 * table fits in 60 % of the device's free memory.  bjj_get_info reports the choice and the bytes.
 * Valid: 0, BJJ_WINDOW_AUTO, 4..28. */
int bjj_init(int device, int window_bits, bjj_ctx** out_ctx);
void bjj_free(bjj_ctx* ctx);
/* Blocks until everything the context has enqueued -- on its own stream and on the callers' streams -- has finished.
 * Returns BJJ_E_HIP if a verify / variable-base workgroup had to give up waiting for a slot of per-lane table scratch since
 * the last synchronising call (it cannot happen unless a kernel was aborted while it held slots): the wait is bounded, the
 * launch ends, and this call reports that the results of those launches are not valid instead of the GPU hanging; the slot
 * queues are rebuilt, the context stays usable.  The host-pointer entry points and the *_multi_dev entry points, which
 * synchronise for the caller, end with the same check and return BJJ_E_HIP themselves. */
int bjj_sync(bjj_ctx* ctx);
/* The context's own stream (hipStream_t as void*). */
void* bjj_stream(bjj_ctx* ctx);

/* ---- pinned host memory ------------------------------------------------------
 * The reference's callers hold their data in ordinary Rust values (Point { x, y }: Fr, BigInt scalars; src/lib.rs:134-138,
 * 149, 239-243, 395): a batch wrapper has to marshal them into the byte records of this boundary anyway, and it should
 * marshal them straight into memory the GPU's copy engines can read -- then the host-pointer entry points below move every
 * byte exactly once (PCIe) instead of staging it through a second buffer.
 *   bjj_host_alloc / bjj_host_free         page-locked memory owned by the library (hipHostMalloc, portable to all devices)
 *   bjj_host_register / bjj_host_unregister pin a range the caller owns (e.g. a Rust Vec<u8> that lives across calls;
 *                                           hipHostRegister: costs ~60 us per MB once, pays off from the second call on)
 *   bjj_host_is_pinned                     1 / 0: would a host-pointer call copy this range directly?
 * Ranges are process-wide (any context may use them); free / unregister them before the last context is released. */
int bjj_host_alloc(bjj_ctx* ctx, size_t bytes, void** out_ptr);
int bjj_host_free(bjj_ctx* ctx, void* ptr);
int bjj_host_register(bjj_ctx* ctx, void* ptr, size_t bytes);
int bjj_host_unregister(bjj_ctx* ctx, void* ptr);
int bjj_host_is_pinned(bjj_ctx* ctx, const void* ptr, size_t bytes);

/* ---- host-pointer batch API ------------------------------------------------ */
int bjj_mul_fixed_base(bjj_ctx* ctx, const uint8_t* scalars /* n*32 */, size_t n,
                       uint8_t* out_xy /* n*64 */);
int bjj_mul_var_base(bjj_ctx* ctx, const uint8_t* pts_xy /* n*64 */,
                     const uint8_t* scalars /* n*32 */, size_t n, uint8_t* out_xy /* n*64 */);
/* scalars: n records of scalar_bytes (a multiple of 32, <= BJJ_MAX_SCALAR_BYTES) little-endian bytes each.  On-curve
 * points use n mod 8l (exact: the group order); off-curve points replay the reference's loop over all n.bits() bits. */
int bjj_mul_var_base_wide(bjj_ctx* ctx, const uint8_t* pts_xy /* n*64 */, const uint8_t* scalars /* n*scalar_bytes */,
                          size_t scalar_bytes, size_t n, uint8_t* out_xy /* n*64 */);
int bjj_poseidon5(bjj_ctx* ctx, const uint8_t* in /* n*160 */, size_t n, uint8_t* out /* n*32 */);
int bjj_eddsa_verify(bjj_ctx* ctx, const uint8_t* pk_xy /* n*64 */, const uint8_t* r_xy /* n*64 */,
                     const uint8_t* s /* n*32 */, const uint8_t* msg /* n*32 */, size_t n,
                     uint8_t* ok /* n */);
/* Schnorr variant: ok[i] = 1 / 0 for Ok(true) / Ok(false), 2 for Err ("msg outside the Finite Field").
 * `s` is a 32-byte integer; the crate's unreduced k + x*h may be wider -- reduce it mod 8l first (exact:
 * it only multiplies the on-curve generator B8). */
int bjj_schnorr_verify(bjj_ctx* ctx, const uint8_t* pk_xy /* n*64 */, const uint8_t* r_xy /* n*64 */,
                       const uint8_t* s /* n*32 */, const uint8_t* msg /* n*32 */, size_t n, uint8_t* ok /* n */);
int bjj_point_add(bjj_ctx* ctx, const uint8_t* p_xy /* n*64 */, const uint8_t* q_xy /* n*64 */,
                  size_t n, uint8_t* out_xy /* n*64 */);
/* Raw PointProjective::add (src/lib.rs:88-131): 96-byte (x, y, z) records, any z, result not normalised -- the
 * canonical values of the three field elements the reference computes.  bjj_proj_affine: src/lib.rs:70-85. */
int bjj_proj_add(bjj_ctx* ctx, const uint8_t* p_xyz /* n*96 */, const uint8_t* q_xyz /* n*96 */, size_t n,
                 uint8_t* out_xyz /* n*96 */);
int bjj_proj_affine(bjj_ctx* ctx, const uint8_t* p_xyz /* n*96 */, size_t n, uint8_t* out_xy /* n*64 */);
/* Wire format (src/lib.rs:166-178): 32 bytes = y little-endian, bit 255 = (x > (r-1)/2). */
int bjj_compress_points(bjj_ctx* ctx, const uint8_t* pts_xy /* n*64 */, size_t n, uint8_t* out /* n*32 */);
/* B8.mul_scalar(n).compress() in one pass: byte-identical to bjj_compress_points(bjj_mul_fixed_base(..)), half the bytes
 * back across PCIe (the copy-out is what bounds the affine form on host pointers) and no second call. */
int bjj_mul_fixed_base_compressed(bjj_ctx* ctx, const uint8_t* scalars /* n*32 */, size_t n, uint8_t* out /* n*32 */);
/* ok[i] = 1 where decompress_point returns Ok, 0 where it returns Err (y >= r, x^2 a non-residue,
 * or x^2 == 0 -- the reference's modsqrt rejects 0); out_xy[i] is all-zero for Err. */
int bjj_decompress_points(bjj_ctx* ctx, const uint8_t* in /* n*32 */, size_t n, uint8_t* out_xy /* n*64 */,
                          uint8_t* ok /* n */);
/* pk: compressed point; sig: compressed R (32) then s (32), i.e. Signature::compress()'s layout
 * (src/lib.rs:245-258).  ok[i] = 1 / 0 as bjj_eddsa_verify, or 2 where pk or R fails to
 * decompress (the reference returns Err before verify is reached). */
int bjj_eddsa_verify_compressed(bjj_ctx* ctx, const uint8_t* pk /* n*32 */, const uint8_t* sig /* n*64 */,
                                const uint8_t* msg /* n*32 */, size_t n, uint8_t* ok /* n */);
/* Signer side.  keys: the 32-byte PrivateKey.key; msg as for verify.  bjj_sign: ok[i] = 0 where the
 * reference returns Err ("msg outside the Finite Field", msg > Q, src/lib.rs:309-311; outputs zeroed);
 * out_s is the canonical integer s < l that Signature.s holds. */
int bjj_scalar_keys(bjj_ctx* ctx, const uint8_t* keys /* n*32 */, size_t n, uint8_t* out /* n*32 */);
/* Signer hardening (off by default; the reference itself branches on secret bits, src/lib.rs:158).  By default the signer
 * entry points (bjj_public_keys, bjj_sign, bjj_sign_schnorr and their _dev forms) index the context's fixed-base table with
 * digits of the nonce and of the scalar key: gather addresses -- and with them cache / TLB / HBM-channel timing -- depend on
 * secrets.  on = 1 switches them to a form in which NO memory address and NO branch depends on a secret: the two fixed-base
 * multiplications run over a separate 4-bit-window table (73 KB, built on first use), every lane reads all nine entries
 * of each window and selects by arithmetic; Blake-512, the mod-l arithmetic, the fixed-trip-count inversion and Poseidon
 * (public inputs) were free of secret-dependent addresses already.  Results are bit-identical; cost: 62 additions instead of
 * 8 per multiplication (bjj_sign about 2x slower).  The verifier entry points handle public data only and are unaffected. */
int bjj_set_signer_constant_time(bjj_ctx* ctx, int on);
int bjj_public_keys(bjj_ctx* ctx, const uint8_t* keys /* n*32 */, size_t n, uint8_t* out_xy /* n*64 */);
/* sk.public().compress() -- what a key server ships (src/lib.rs:304-306, 166-178) */
int bjj_public_keys_compressed(bjj_ctx* ctx, const uint8_t* keys /* n*32 */, size_t n, uint8_t* out /* n*32 */);
int bjj_sign(bjj_ctx* ctx, const uint8_t* keys /* n*32 */, const uint8_t* msgs /* n*32 */, size_t n,
             uint8_t* out_r_xy /* n*64 */, uint8_t* out_s /* n*32 */, uint8_t* ok /* n */);
/* sk.sign(msg)?.compress(): 64 bytes per signature = Point::compress(R) then s as 32 little-endian bytes
 * (Signature::compress, src/lib.rs:245-258); all-zero with ok[i] = 0 where the reference returns Err. */
int bjj_sign_compressed(bjj_ctx* ctx, const uint8_t* keys /* n*32 */, const uint8_t* msgs /* n*32 */, size_t n,
                        uint8_t* out_sig /* n*64 */, uint8_t* ok /* n */);
/* PrivateKey::sign_schnorr(m) -> Result<(Point, BigInt), String> (src/lib.rs:344-361) with the nonce supplied by
 * the caller: the reference draws k = rng.gen_biguint(1024) (:347-348) -- randomness stays on the host, so the
 * call is deterministic.  nonces: 128-byte little-endian integers.  out_r = k*B8; out_s = k + scalar_key*h as
 * the reference's UNREDUCED integer (:359, < 2^1025) in a 160-byte little-endian record (reduce it mod 8l
 * before bjj_schnorr_verify).  ok[i] = 0 where the reference returns Err (msg > Q, :365-367; outputs zeroed). */
#define BJJ_SCHNORR_NONCE_BYTES 128
#define BJJ_SCHNORR_S_BYTES 160
int bjj_sign_schnorr(bjj_ctx* ctx, const uint8_t* keys /* n*32 */, const uint8_t* msgs /* n*32 */,
                     const uint8_t* nonces /* n*128 */, size_t n, uint8_t* out_r_xy /* n*64 */,
                     uint8_t* out_s /* n*160 */, uint8_t* ok /* n */);

/* ---- device-pointer batch API (asynchronous on `stream`) -------------------- */
int bjj_mul_fixed_base_dev(bjj_ctx* ctx, const void* d_scalars, size_t n, void* d_out_xy, void* stream);
int bjj_mul_var_base_dev(bjj_ctx* ctx, const void* d_pts_xy, const void* d_scalars, size_t n,
                         void* d_out_xy, void* stream);
int bjj_mul_var_base_wide_dev(bjj_ctx* ctx, const void* d_pts_xy, const void* d_scalars, size_t scalar_bytes, size_t n,
                              void* d_out_xy, void* stream);
int bjj_poseidon5_dev(bjj_ctx* ctx, const void* d_in, size_t n, void* d_out, void* stream);
int bjj_eddsa_verify_dev(bjj_ctx* ctx, const void* d_pk_xy, const void* d_r_xy, const void* d_s,
                         const void* d_msg, size_t n, void* d_ok, void* stream);
int bjj_schnorr_verify_dev(bjj_ctx* ctx, const void* d_pk_xy, const void* d_r_xy, const void* d_s,
                           const void* d_msg, size_t n, void* d_ok, void* stream);
int bjj_point_add_dev(bjj_ctx* ctx, const void* d_p_xy, const void* d_q_xy, size_t n, void* d_out_xy,
                      void* stream);
int bjj_proj_add_dev(bjj_ctx* ctx, const void* d_p_xyz, const void* d_q_xyz, size_t n, void* d_out_xyz, void* stream);
int bjj_proj_affine_dev(bjj_ctx* ctx, const void* d_p_xyz, size_t n, void* d_out_xy, void* stream);
int bjj_scalar_keys_dev(bjj_ctx* ctx, const void* d_keys, size_t n, void* d_out, void* stream);
int bjj_public_keys_dev(bjj_ctx* ctx, const void* d_keys, size_t n, void* d_out_xy, void* stream);
int bjj_sign_dev(bjj_ctx* ctx, const void* d_keys, const void* d_msgs, size_t n, void* d_out_r_xy, void* d_out_s,
                 void* d_ok, void* stream);
int bjj_sign_schnorr_dev(bjj_ctx* ctx, const void* d_keys, const void* d_msgs, const void* d_nonces, size_t n,
                         void* d_out_r_xy, void* d_out_s, void* d_ok, void* stream);
int bjj_compress_points_dev(bjj_ctx* ctx, const void* d_pts_xy, size_t n, void* d_out, void* stream);
int bjj_decompress_points_dev(bjj_ctx* ctx, const void* d_in, size_t n, void* d_out_xy, void* d_ok, void* stream);
int bjj_eddsa_verify_compressed_dev(bjj_ctx* ctx, const void* d_pk, const void* d_sig, const void* d_msg, size_t n,
                                    void* d_ok, void* stream);
int bjj_mul_fixed_base_compressed_dev(bjj_ctx* ctx, const void* d_scalars, size_t n, void* d_out /* n*32 */, void* stream);
int bjj_public_keys_compressed_dev(bjj_ctx* ctx, const void* d_keys, size_t n, void* d_out /* n*32 */, void* stream);
int bjj_sign_compressed_dev(bjj_ctx* ctx, const void* d_keys, const void* d_msgs, size_t n, void* d_out_sig /* n*64 */,
                            void* d_ok, void* stream);

/* Makes sure the context's scratch can serve batches of up to n items, so that
 * later *_dev calls do not allocate (call once before timing). */
int bjj_reserve(bjj_ctx* ctx, size_t n);

/* Verifies the whole fixed-base table on the device by induction over its entries
 * (T[j][0] = O, T[j][k] + P_j = T[j][k+1], P_j+1 = 2 T[j][2^(W-1)], P_0 = B8, canonical
 * limbs): *n_bad = number of violated conditions (0 for a sound table). */
int bjj_check_table(bjj_ctx* ctx, uint64_t* n_bad);

/* Introspection for benchmarks / profiling reports.  The caller sets struct_size = sizeof(bjj_info) of ITS build before
 * the call; the library fills at most that many bytes, so the struct can grow at its end without overrunning a caller that
 * was compiled against an older header (a struct_size below 8 -- e.g. an uninitialised struct -- is BJJ_E_INVALID). */
typedef struct {
  uint32_t struct_size;     /* IN: sizeof(bjj_info) as the caller knows it */
  int device;
  int compute_units;
  int window_bits;          /* fixed-base W */
  int n_windows;            /* ceil(252 / W) */
  uint64_t table_bytes;     /* fixed-base table size in HBM */
  uint64_t scratch_bytes;   /* current scratch allocation */
  const char* kernel_fixed_base; /* kernel symbol names, for matching rocprofv3 rows */
  const char* kernel_var_base;
  const char* kernel_poseidon5;
  const char* kernel_verify;
  double init_ms;           /* wall time of bjj_init (allocation + table build on the GPU) */
  int signer_constant_time; /* bjj_set_signer_constant_time: 1 = the signer entry points scan a small table (see there) */
  /* since 0.5.0 -- what the context's LAST calls did (-1 = no such call yet); for tests, profiles and the bench line */
  int last_fixed_base_shape;   /* 0 = one 512-lane workgroup per CU (a launch that runs alone), 1 = two of 256 (overlapping launches), 2 = four lanes per item (short calls, since 0.6.0) */
  int last_var_base_form;      /* 1 = tiles (alone), 0 = grid-strided (overlapping), 2 = four lanes per item (short calls, since 0.6.0) */
  int last_verify_dispatch;    /* 0 = persistent waves (one launch > 2^21 items that runs alone), 1 = one group per workgroup, 2 = eight lanes per signature (short calls, since 0.6.0) */
  uint32_t last_host_direct_arrays;  /* last host-pointer call: arrays copied straight from / to pinned caller memory ... */
  uint32_t last_host_staged_arrays;  /* ... arrays staged through the context's pinned buffers (pageable caller memory) ... */
  uint32_t last_host_chunks;         /* ... and the chunks it was cut into */
  int host_copy_threads;       /* copy workers of the staged path (0 until a pageable array has been seen) */
  const char* kernel_fixed_base_overlap; /* kernel symbols of the forms overlapping launches get (two streams) */
  const char* kernel_var_base_overlap;
  /* since 0.6.0 */
  int last_var_base_split;     /* variable base, the exact kernel for off-curve points: 0 = behind the batch kernel, 1 = beside it (scan first) */
  uint32_t last_host_zero_copy; /* last host-pointer call, bit 0: the kernels stored their results into the (pinned) output array themselves, no copy-out stage;
                                   bit 1: the kernels read the (pinned) input arrays themselves, no copy-in stage */
  int last_poseidon_form;      /* 0 = one hash per lane, 1 = six lanes per hash (short calls) */
  int last_sign_form;          /* bjj_sign / bjj_sign_compressed: 0 = one signature per lane, 1 = eight lanes per signature (short calls) */
} bjj_info;
int bjj_get_info(bjj_ctx* ctx, bjj_info* info);

/* ---- multi-GPU (BASELINE.json configs[4]) -----------------------------------------
 * The path shards with no exchange step (src/lib.rs:149-164, 395-412 have no cross-item state): a
 * bjj_multi handle owns one context per device (replicated fixed-base table), rank i of G processes
 * the contiguous block [i*ceil(n/G), min(n, (i+1)*ceil(n/G))) -- bjj_shard_bounds.  One process,
 * all devices: this is the form a Rust host calls.
 *   host-pointer form    arrays in host memory; one host thread per device drives that device's
 *                        host-pointer pipeline over its block (pinned arrays copied directly).  No inter-GPU traffic.
 *   *_multi_dev form     arrays resident in the HBM of the handle's FIRST device (rank 0 holds all
 *                        inputs, as in BASELINE cfg 5).  Transfers over xGMI and kernels are pipelined:
 *                        the root's block is processed in place from t = 0 while its sends run on a
 *                        separate transfer stream; every peer's block travels in up to `chunks` pieces
 *                        (bjj_multi_set_chunks), piece c of all peers and all input arrays in ONE
 *                        ncclGroup of exact-count ncclSend / ncclRecv pairs; a peer computes piece c
 *                        as soon as it has arrived, while piece c+1 is in flight, and returns its
 *                        results piece by piece.  chunks = 1 with G | n is the serial schedule: one
 *                        grouped ncclScatter (in place at the root), the kernels, one grouped
 *                        ncclGather.  Synchronous; the input arrays must be complete when the call is
 *                        made (no pending writes on other streams).  RCCL is loaded with dlopen on the
 *                        first such call (single process, ncclCommInitAll; "librccl.so.1", or the
 *                        library named by the environment variable BJJ_RCCL_LIBRARY); BJJ_E_RCCL if
 *                        absent.  If a transfer fails after the call has started to enqueue, the handle
 *                        aborts its communicators and becomes unusable (every later *_multi_dev call
 *                        returns BJJ_E_INVALID): release it with bjj_multi_free. */
typedef struct bjj_multi bjj_multi;
/* devices: n_devices HIP device indices (NULL = 0 .. n_devices-1; NULL and 0 = all visible devices).  A device may be
 * named more than once (several contexts on one GPU; how the G > 1 block arithmetic is tested on a one-GPU box) -- the
 * host-pointer form and BJJ_TRANSPORT_PEER_COPY accept that, RCCL does not (BJJ_E_RCCL). */
int bjj_multi_init(const int* devices, int n_devices, int window_bits, bjj_multi** out);
/* Transport of the *_multi_dev form: BJJ_TRANSPORT_RCCL (default: grouped ncclSend / ncclRecv, resp. ncclScatter /
 * ncclGather, over xGMI) or BJJ_TRANSPORT_PEER_COPY (hipMemcpyPeerAsync of the same pieces on the peers' transfer
 * streams: copy engines, no compute units, no RCCL). */
enum { BJJ_TRANSPORT_RCCL = 0, BJJ_TRANSPORT_PEER_COPY = 1 };
int bjj_multi_set_transport(bjj_multi* m, int transport);
/* Pipeline depth of the *_multi_dev form: a peer's block is cut into at most `chunks` pieces (1..16; default 4, or the
 * environment variable BJJ_MULTI_CHUNKS at bjj_multi_init) of at least min_chunk_items items each (default 32768; pieces
 * are multiples of 64 items).  chunks = 1 restores the serial scatter -> kernels -> gather schedule. */
int bjj_multi_set_chunks(bjj_multi* m, int chunks, size_t min_chunk_items);
void bjj_multi_free(bjj_multi* m);
int bjj_multi_size(const bjj_multi* m);
bjj_ctx* bjj_multi_ctx(bjj_multi* m, int rank);     /* the per-device context (owned by the handle) */
int bjj_multi_device(const bjj_multi* m, int rank); /* HIP device index of a rank */
void bjj_shard_bounds(size_t n, int n_devices, int rank, size_t* lo, size_t* hi);
int bjj_mul_fixed_base_multi(bjj_multi* m, const uint8_t* scalars /* n*32 */, size_t n, uint8_t* out_xy /* n*64 */);
int bjj_mul_var_base_multi(bjj_multi* m, const uint8_t* pts_xy /* n*64 */, const uint8_t* scalars /* n*32 */, size_t n,
                           uint8_t* out_xy /* n*64 */);
int bjj_eddsa_verify_multi(bjj_multi* m, const uint8_t* pk_xy, const uint8_t* r_xy, const uint8_t* s, const uint8_t* msg,
                           size_t n, uint8_t* ok /* n */);
int bjj_mul_fixed_base_multi_dev(bjj_multi* m, const void* d_scalars, size_t n, void* d_out_xy);
int bjj_mul_var_base_multi_dev(bjj_multi* m, const void* d_pts_xy, const void* d_scalars, size_t n, void* d_out_xy);
int bjj_eddsa_verify_multi_dev(bjj_multi* m, const void* d_pk_xy, const void* d_r_xy, const void* d_s, const void* d_msg,
                               size_t n, void* d_ok /* 16-byte aligned */);
/* Phase spans of the last *_multi_dev call (HIP events, each device against its own start; maximum over the devices):
 * scatter = until a peer's last input piece has arrived, compute = first kernel start to last kernel end on a device,
 * gather = last kernel end to last result piece delivered; and the RCCL version in use.  In the pipelined schedule
 * the spans overlap; their sum is what the serial schedule costs. */
int bjj_multi_last_timing(bjj_multi* m, double* scatter_ms, double* compute_ms, double* gather_ms, int* rccl_version);
/* ... and what the call took as a whole: total_ms = start to last event on any device (HIP events), wall_ms = host
 * clock around the call, chunks = pieces per peer block actually used.  Compare total_ms with scatter + compute + gather
 * of bjj_multi_last_timing, or with total_ms of the same call after bjj_multi_set_chunks(m, 1, ..). */
int bjj_multi_last_overlap(bjj_multi* m, double* total_ms, double* wall_ms, int* chunks);

#ifdef __cplusplus
}
#endif
#endif /* BJJ_HIP_H */
