//! `babyjubjub_rs` with the hot path on an AMD MI355X.
//!
//! The public items below keep the names and signatures of arnaucube/babyjubjub-rs v0.0.11 (`src/lib.rs` of the
//! reference, line numbers in the comments) for everything on the accelerated path; the arithmetic happens in
//! libbjj_hip.so (`../include/bjj_hip.h`).  What the reference does not have are the `*_batch` functions at the end --
//! the single-item calls cost one GPU launch each and exist for source compatibility.
//!
//! NOT COMPILED in this repository's build image (no Rust toolchain there): see README.md.
extern crate ff;
#[macro_use]
extern crate lazy_static;
extern crate num_bigint;
extern crate num_traits;
extern crate poseidon_rs;
extern crate rand;

pub mod ffi;
pub mod gpu;
pub mod multi;
pub mod utils; // lib.rs:24

use ff::{Field, PrimeField};
use gpu::with_gpu;
use num_bigint::{BigInt, RandBigInt, Sign};
use num_traits::Zero;

pub type Fr = poseidon_rs::Fr; // lib.rs:7

// ---------------------------------------------------------------------------------------------------------
// marshalling between the crate's types and the 32-byte little-endian records of the C ABI
// ---------------------------------------------------------------------------------------------------------
fn fr_to_bytes(f: &Fr) -> [u8; 32] {
    let repr = f.into_repr(); // canonical value, [u64; 4] little-endian limbs
    let mut out = [0u8; 32];
    for (i, limb) in repr.as_ref().iter().enumerate() {
        out[8 * i..8 * i + 8].copy_from_slice(&limb.to_le_bytes());
    }
    out
}

fn fr_from_bytes(b: &[u8]) -> Fr {
    let mut repr = <Fr as PrimeField>::Repr::default();
    for (i, limb) in repr.as_mut().iter_mut().enumerate() {
        let mut w = [0u8; 8];
        w.copy_from_slice(&b[8 * i..8 * i + 8]);
        *limb = u64::from_le_bytes(w);
    }
    Fr::from_repr(repr).expect("libbjj_hip returns canonical field elements")
}

/// `BigInt::to_bytes_le` zero-padded to `width` bytes (sign dropped, as lib.rs:156 / lib.rs:249-252 do)
fn bigint_to_le(n: &BigInt, width: usize) -> Vec<u8> {
    let (_, mut b) = n.to_bytes_le();
    b.resize(width, 0);
    b
}

fn bigint_from_le(b: &[u8]) -> BigInt {
    BigInt::from_bytes_le(Sign::Plus, b)
}

lazy_static! {
    /// The field modulus r -- the reference's one public constant (lib.rs:33-36; D, A, B8, ORDER, SUBORDER are private there).
    pub static ref Q: BigInt =
        BigInt::parse_bytes(b"21888242871839275222246405745257275088548364400416034343698204186575808495617", 10).unwrap();
    /// group order 8 l (lib.rs:48-52)
    static ref ORDER: BigInt =
        BigInt::parse_bytes(b"21888242871839275222246405745257275088614511777268538073601725287587578984328", 10).unwrap();
    /// generator of the prime-order subgroup (lib.rs:37-46)
    static ref B8: Point = Point {
        x: Fr::from_str("5299619240641551281634865583518297030282874472190772894086521144482721001553").unwrap(),
        y: Fr::from_str("16950150798460657717958625567821834550301663161624707787222815936182638968203").unwrap(),
    };
}

/// `Fr::from_str(&msg.to_string()).unwrap()` of the reference (lib.rs:321, 368, 399) panics on a negative integer -- the
/// range test before it only rejects msg > Q.  The 32-byte records of the C ABI cannot carry a sign, so the wrappers
/// reproduce the panic here instead of silently verifying / signing |msg|.
fn non_negative(msg: &BigInt) {
    if msg.sign() == Sign::Minus {
        panic!("called `Option::unwrap()` on a `None` value: Fr::from_str of a negative msg");
    }
}

/// |n| mod 8l: what `B8.mul_scalar(&n)` depends on (the reference drops the sign, lib.rs:156; B8 has order dividing 8l)
fn b8_scalar(n: &BigInt) -> BigInt {
    let (_, mag) = n.clone().into_parts();
    BigInt::from(mag) % &*ORDER
}

fn point_bytes(p: &Point) -> [u8; 64] {
    let mut out = [0u8; 64];
    out[..32].copy_from_slice(&fr_to_bytes(&p.x));
    out[32..].copy_from_slice(&fr_to_bytes(&p.y));
    out
}

fn point_from_bytes(b: &[u8]) -> Point {
    Point { x: fr_from_bytes(&b[..32]), y: fr_from_bytes(&b[32..64]) }
}

fn proj_bytes(p: &PointProjective) -> [u8; 96] {
    let mut out = [0u8; 96];
    out[..32].copy_from_slice(&fr_to_bytes(&p.x));
    out[32..64].copy_from_slice(&fr_to_bytes(&p.y));
    out[64..].copy_from_slice(&fr_to_bytes(&p.z));
    out
}

// ---------------------------------------------------------------------------------------------------------
// PointProjective (lib.rs:62-132)
// ---------------------------------------------------------------------------------------------------------
#[derive(Clone, Debug)]
pub struct PointProjective {
    pub x: Fr,
    pub y: Fr,
    pub z: Fr,
}

impl PointProjective {
    pub fn affine(&self) -> Point {
        // lib.rs:70-85 (z == 0 -> (0, 0))
        let out = with_gpu(|g| g.proj_affine(&proj_bytes(self))).expect("bjj_proj_affine");
        point_from_bytes(&out)
    }

    #[allow(clippy::many_single_char_names)]
    pub fn add(&self, q: &PointProjective) -> PointProjective {
        // lib.rs:88-131: the raw (x, y, z) of the reference's formula sequence, any z
        let out = with_gpu(|g| g.proj_add(&proj_bytes(self), &proj_bytes(q))).expect("bjj_proj_add");
        PointProjective { x: fr_from_bytes(&out[..32]), y: fr_from_bytes(&out[32..64]), z: fr_from_bytes(&out[64..96]) }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Point (lib.rs:134-186)
// ---------------------------------------------------------------------------------------------------------
#[derive(Clone, Debug)]
pub struct Point {
    pub x: Fr,
    pub y: Fr,
}

impl Point {
    pub fn projective(&self) -> PointProjective {
        // lib.rs:141-147
        PointProjective { x: self.x, y: self.y, z: Fr::one() }
    }

    /// COST OF THE n = 1 CALL: about 0.5 ms (a kernel launch, two PCIe round trips and the multiplication spread over four lanes; 1.2 ms
    /// until libbjj_hip 0.6.0) against 0.17 ms for the reference on one CPU core -- item by item this drop-in is SLOWER than the crate
    /// it replaces; the GPU pays from 3 items per call on.
    /// Use `mul_scalar_batch` / `mul_fixed_base_batch` (first table).
    pub fn mul_scalar(&self, n: &BigInt) -> Point {
        // lib.rs:149-164: abs(n) * P, n of any size, not reduced
        mul_scalar_batch(std::slice::from_ref(self), std::slice::from_ref(n)).pop().unwrap()
    }

    pub fn compress(&self) -> [u8; 32] {
        // lib.rs:166-178
        let out = with_gpu(|g| g.compress_points(&point_bytes(self))).expect("bjj_compress_points");
        let mut r = [0u8; 32];
        r.copy_from_slice(&out);
        r
    }

    pub fn equals(&self, p: Point) -> bool {
        // lib.rs:180-185
        self.x == p.x && self.y == p.y
    }
}

pub fn test_bit(b: &[u8], i: usize) -> bool {
    b[i / 8] & (1 << (i % 8)) != 0 // lib.rs:188-190
}

pub fn decompress_point(bb: [u8; 32]) -> Result<Point, String> {
    // lib.rs:192-224
    let (pts, ok) = with_gpu(|g| g.decompress_points(&bb))?;
    if ok[0] == 0 {
        return Err("y outside the Finite Field over R, or x^2 not a non-zero square".to_string());
    }
    Ok(point_from_bytes(&pts))
}

// ---------------------------------------------------------------------------------------------------------
// Signature (lib.rs:239-268)
// ---------------------------------------------------------------------------------------------------------
#[derive(Debug, Clone)]
pub struct Signature {
    pub r_b8: Point,
    pub s: BigInt,
}

impl Signature {
    pub fn compress(&self) -> [u8; 64] {
        // lib.rs:246-258
        let mut b = [0u8; 64];
        b[..32].copy_from_slice(&self.r_b8.compress());
        b[32..].copy_from_slice(&bigint_to_le(&self.s, 32)[..32]);
        b
    }
}

pub fn decompress_signature(b: &[u8; 64]) -> Result<Signature, String> {
    // lib.rs:260-268
    let mut r = [0u8; 32];
    r.copy_from_slice(&b[..32]);
    Ok(Signature { r_b8: decompress_point(r)?, s: bigint_from_le(&b[32..]) })
}

// ---------------------------------------------------------------------------------------------------------
// PrivateKey (lib.rs:270-361)
// ---------------------------------------------------------------------------------------------------------
pub struct PrivateKey {
    pub key: [u8; 32],
}

impl PrivateKey {
    pub fn import(b: Vec<u8>) -> Result<PrivateKey, String> {
        // lib.rs:275-282
        if b.len() != 32 {
            return Err(String::from("imported key can not be bigger than 32 bytes"));
        }
        let mut key = [0u8; 32];
        key.copy_from_slice(&b);
        Ok(PrivateKey { key })
    }

    pub fn scalar_key(&self) -> BigInt {
        // lib.rs:284-302: Blake-512, prune, >> 3
        bigint_from_le(&with_gpu(|g| g.scalar_keys(&self.key)).expect("bjj_scalar_keys"))
    }

    pub fn public(&self) -> Point {
        // lib.rs:304-306
        point_from_bytes(&with_gpu(|g| g.public_keys(&self.key)).expect("bjj_public_keys"))
    }

    pub fn sign(&self, msg: BigInt) -> Result<Signature, String> {
        // lib.rs:308-342
        if msg > *Q {
            return Err("msg outside the Finite Field".to_string());
        }
        non_negative(&msg); // lib.rs:321
        let (r, s, ok) = with_gpu(|g| g.sign(&self.key, &bigint_to_le(&msg, 32)))?;
        if ok[0] == 0 {
            return Err("msg outside the Finite Field".to_string());
        }
        Ok(Signature { r_b8: point_from_bytes(&r), s: bigint_from_le(&s) })
    }

    pub fn sign_schnorr(&self, m: BigInt) -> Result<(Point, BigInt), String> {
        // lib.rs:344-361; the 1024-bit nonce is drawn here (lib.rs:347-348) and handed to the device
        let k = BigInt::from(rand::thread_rng().gen_biguint(1024)); // drawn first, as at lib.rs:347-348
        if m > *Q {
            return Err("msg outside the Finite Field".to_string()); // schnorr_hash's Err, lib.rs:365-367 via :353
        }
        non_negative(&m); // lib.rs:368
        let (r, s, ok) = with_gpu(|g| g.sign_schnorr(&self.key, &bigint_to_le(&m, 32), &bigint_to_le(&k, ffi::BJJ_SCHNORR_NONCE_BYTES)))?;
        if ok[0] == 0 {
            return Err("msg outside the Finite Field".to_string());
        }
        Ok((point_from_bytes(&r), bigint_from_le(&s))) // s = k + scalar_key * h, unreduced (lib.rs:359)
    }
}

pub fn schnorr_hash(pk: &Point, msg: BigInt, c: &Point) -> Result<BigInt, String> {
    // lib.rs:364-373: Poseidon([pk.x, pk.y, c.x, c.y, msg])
    if msg > *Q {
        return Err("msg outside the Finite Field".to_string());
    }
    non_negative(&msg); // lib.rs:368
    let mut input = Vec::with_capacity(160);
    input.extend_from_slice(&point_bytes(pk));
    input.extend_from_slice(&point_bytes(c));
    input.extend_from_slice(&bigint_to_le(&msg, 32));
    Ok(bigint_from_le(&with_gpu(|g| g.poseidon5(&input))?))
}

pub fn verify_schnorr(pk: Point, m: BigInt, r: Point, s: BigInt) -> Result<bool, String> {
    // lib.rs:375-385; s only multiplies B8, so it is reduced mod the group order 8l into the 32-byte record (exact)
    // sl = B8.mul_scalar(&s) comes first in the reference (:377) and uses |s|; the hash's Err / panic follow (:379)
    if m > *Q {
        return Err("msg outside the Finite Field".to_string());
    }
    non_negative(&m);
    let s_red = b8_scalar(&s);
    let ok = with_gpu(|g| g.schnorr_verify(&point_bytes(&pk), &point_bytes(&r), &bigint_to_le(&s_red, 32), &bigint_to_le(&m, 32)))?;
    match ok[0] {
        2 => Err("msg outside the Finite Field".to_string()),
        v => Ok(v == 1),
    }
}

pub fn new_key() -> PrivateKey {
    // lib.rs:387-393: 1024 random bits; the first 32 bytes of their big-endian form become the key
    let raw = rand::thread_rng().gen_biguint(1024);
    let be = raw.to_bytes_be();
    PrivateKey::import(be[..32].to_vec()).unwrap()
}

/// COST OF THE n = 1 CALL: about 0.6 ms (eight lanes per signature; 1.4 ms until libbjj_hip 0.6.0) against 0.46 ms for the reference on one
/// CPU core -- a caller that verifies signature by signature gets slower; break-even is 2 signatures per call.  Use `verify_batch`
/// (first table).
pub fn verify(pk: Point, sig: Signature, msg: BigInt) -> bool {
    // lib.rs:395-412
    if msg > *Q {
        return false; // :396-398
    }
    non_negative(&msg); // :399
    verify_batch(&[pk], &[sig], &[msg])[0]
}

// ---------------------------------------------------------------------------------------------------------
// batch forms (not in the reference): one call, one pass over the whole slice on the GPU
// ---------------------------------------------------------------------------------------------------------
/// `points[i].mul_scalar(&scalars[i])` for all i.  Scalars of any size: records are as wide as the longest scalar
/// (rounded up to 32 bytes); points equal to B8 go through the fixed-base engine when ALL of them are B8.
pub fn mul_scalar_batch(points: &[Point], scalars: &[BigInt]) -> Vec<Point> {
    assert_eq!(points.len(), scalars.len());
    if points.is_empty() {
        return Vec::new();
    }
    let width = scalars.iter().map(|n| ((n.bits() as usize + 255) / 256).max(1) * 32).max().unwrap();
    let all_b8 = points.iter().all(|p| p.x == B8.x && p.y == B8.y);
    let n = points.len();
    // records are written straight into pinned memory: the library then moves every byte once, over PCIe (include/bjj_hip.h)
    with_gpu(|gpu| {
        let mut sc = gpu.pinned(n * width).expect("mul_scalar_batch");
        for (i, k) in scalars.iter().enumerate() {
            sc[i * width..(i + 1) * width].copy_from_slice(&bigint_to_le(k, width));
        }
        let mut out = gpu.pinned(n * 64).expect("mul_scalar_batch");
        if all_b8 && width == 32 {
            gpu.mul_fixed_base_into(&sc, &mut out)
        } else {
            let mut pts = gpu.pinned(n * 64).expect("mul_scalar_batch");
            for (i, p) in points.iter().enumerate() {
                pts[i * 64..(i + 1) * 64].copy_from_slice(&point_bytes(p));
            }
            gpu.mul_var_base_into(&pts, &sc, width, &mut out)
        }
        .expect("mul_scalar_batch");
        out.chunks(64).map(point_from_bytes).collect()
    })
}

/// `B8.mul_scalar(&n)` for every n (the engine of `PrivateKey::public`, lib.rs:304-306)
pub fn mul_fixed_base_batch(scalars: &[BigInt]) -> Vec<Point> {
    let n = scalars.len();
    if n == 0 {
        return Vec::new();
    }
    with_gpu(|g| {
        let mut sc = g.pinned(n * 32).expect("mul_fixed_base_batch");
        for (i, k) in scalars.iter().enumerate() {
            // B8 is on the curve: n * B8 == (|n| mod 8l) * B8 exactly, which also brings any n into the 32-byte record
            sc[i * 32..(i + 1) * 32].copy_from_slice(&bigint_to_le(&b8_scalar(k), 32));
        }
        let mut out = g.pinned(n * 64).expect("mul_fixed_base_batch");
        g.mul_fixed_base_into(&sc, &mut out).expect("mul_fixed_base_batch");
        out.chunks(64).map(point_from_bytes).collect()
    })
}

/// `verify(pks[i], sigs[i], msgs[i])` for all i.  A batch cannot panic for one item: a NEGATIVE msg -- where the
/// reference's single-item `verify` panics (lib.rs:399), and so does `verify` above -- yields `false` here.
pub fn verify_batch(pks: &[Point], sigs: &[Signature], msgs: &[BigInt]) -> Vec<bool> {
    assert!(pks.len() == sigs.len() && sigs.len() == msgs.len());
    let n = pks.len();
    let qq = &*Q;
    if n == 0 {
        return Vec::new();
    }
    with_gpu(|g| {
        let (mut pk, mut r) = (g.pinned(n * 64).expect("verify_batch"), g.pinned(n * 64).expect("verify_batch"));
        let (mut s, mut m) = (g.pinned(n * 32).expect("verify_batch"), g.pinned(n * 32).expect("verify_batch"));
        let mut early_false = vec![false; n];
        for i in 0..n {
            pk[i * 64..(i + 1) * 64].copy_from_slice(&point_bytes(&pks[i]));
            r[i * 64..(i + 1) * 64].copy_from_slice(&point_bytes(&sigs[i].r_b8));
            // msg > Q is `false` before anything else (lib.rs:396-398); s wider than 256 bits only multiplies B8: reduce mod 8l
            early_false[i] = msgs[i] > *qq || msgs[i].sign() == Sign::Minus;
            let msg = if early_false[i] { BigInt::zero() } else { msgs[i].clone() };
            // s only multiplies B8 (:405) and its sign is dropped there: any s, of any width, as |s| mod 8l
            let sv = if sigs[i].s.bits() > 256 { b8_scalar(&sigs[i].s) } else { sigs[i].s.clone() };
            s[i * 32..(i + 1) * 32].copy_from_slice(&bigint_to_le(&sv, 32));
            m[i * 32..(i + 1) * 32].copy_from_slice(&bigint_to_le(&msg, 32));
        }
        let mut ok = g.pinned(n).expect("verify_batch");
        g.eddsa_verify_into(&pk, &r, &s, &m, &mut ok).expect("verify_batch");
        (0..n).map(|i| !early_false[i] && ok[i] == 1).collect()
    })
}

/// `PrivateKey::public` for every key
pub fn public_batch(keys: &[PrivateKey]) -> Vec<Point> {
    let mut kb = Vec::with_capacity(keys.len() * 32);
    for k in keys {
        kb.extend_from_slice(&k.key);
    }
    with_gpu(|g| g.public_keys(&kb)).expect("public_batch").chunks(64).map(point_from_bytes).collect()
}

/// `keys[i].public().compress()` for all i: what a key server ships.  One pass on the GPU (the compression is fused into the
/// multiplication's epilogue, `bjj_public_keys_compressed`): 32 bytes per key come back instead of 64.
pub fn public_compressed_batch(keys: &[PrivateKey]) -> Vec<[u8; 32]> {
    let mut kb = Vec::with_capacity(keys.len() * 32);
    for k in keys {
        kb.extend_from_slice(&k.key);
    }
    with_gpu(|g| g.public_keys_compressed(&kb))
        .expect("public_compressed_batch")
        .chunks(32)
        .map(|c| {
            let mut r = [0u8; 32];
            r.copy_from_slice(c);
            r
        })
        .collect()
}

/// `keys[i].sign(msgs[i]).map(|s| s.compress())` for all i (`bjj_sign_compressed`: 64 bytes per signature)
pub fn sign_compressed_batch(keys: &[PrivateKey], msgs: &[BigInt]) -> Vec<Result<[u8; 64], String>> {
    assert_eq!(keys.len(), msgs.len());
    let qq = &*Q;
    let bad: Vec<bool> = msgs.iter().map(|m| m > qq || m.sign() == Sign::Minus).collect();
    let (mut kb, mut mb) = (Vec::with_capacity(keys.len() * 32), Vec::with_capacity(keys.len() * 32));
    for (i, k) in keys.iter().enumerate() {
        kb.extend_from_slice(&k.key);
        mb.extend_from_slice(&bigint_to_le(if bad[i] { qq } else { &msgs[i] }, 32));
    }
    let (sig, ok) = with_gpu(|g| g.sign_compressed(&kb, &mb)).expect("sign_compressed_batch");
    (0..keys.len())
        .map(|i| {
            if bad[i] || ok[i] == 0 {
                Err("msg outside the Finite Field".to_string())
            } else {
                let mut r = [0u8; 64];
                r.copy_from_slice(&sig[64 * i..64 * i + 64]);
                Ok(r)
            }
        })
        .collect()
}

/// `keys[i].sign(msgs[i])` for all i
pub fn sign_batch(keys: &[PrivateKey], msgs: &[BigInt]) -> Vec<Result<Signature, String>> {
    assert_eq!(keys.len(), msgs.len());
    let qq = &*Q;
    // msg > Q is the reference's Err; a negative msg (its panic, lib.rs:321) becomes an Err of this item as well
    let bad: Vec<bool> = msgs.iter().map(|m| m > qq || m.sign() == Sign::Minus).collect();
    let (mut kb, mut mb) = (Vec::with_capacity(keys.len() * 32), Vec::with_capacity(keys.len() * 32));
    for (i, k) in keys.iter().enumerate() {
        kb.extend_from_slice(&k.key);
        mb.extend_from_slice(&bigint_to_le(if bad[i] { qq } else { &msgs[i] }, 32));
    }
    let (r, s, ok) = with_gpu(|g| g.sign(&kb, &mb)).expect("sign_batch");
    (0..keys.len())
        .map(|i| {
            if bad[i] || ok[i] == 0 {
                Err("msg outside the Finite Field".to_string())
            } else {
                Ok(Signature { r_b8: point_from_bytes(&r[64 * i..64 * i + 64]), s: bigint_from_le(&s[32 * i..32 * i + 32]) })
            }
        })
        .collect()
}
