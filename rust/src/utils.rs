//! `babyjubjub_rs::utils` -- the crate's public big-integer helpers (reference `src/utils.rs`: `modulus` :7,
//! `modinv` :11, `concatenate_arrays` :104, `modsqrt` :109, `modsqrt_v2` :164, `legendre_symbol` :215), kept so that
//! `use babyjubjub_rs::utils::*;` in downstream code keeps compiling.  They are host-side number theory on `BigInt`s
//! and never touch the GPU: the accelerated path carries its own field arithmetic (`decompress_point` takes its square
//! root on the device).  Written against the published algorithms (extended Euclid, Euler's criterion, Tonelli-Shanks)
//! with the reference's signatures, `Err` strings and conventions: residues are returned in `[0, q)`,
//! `legendre_symbol` answers 1 for everything that is not a non-residue (0 included), and Tonelli-Shanks starts from
//! the least non-residue `n >= 2`, which fixes WHICH of the two roots comes out (pinned by the reference's own vector,
//! `src/utils.rs:238-258`, and by `tests/test_rust_shim.py` through a model of this file).
//!
//! NOT COMPILED in this repository's build image (no Rust toolchain there): see README.md.
use num_bigint::{BigInt, Sign};
use num_traits::{One, Zero};

/// `a mod m` with the sign of `m` (non-negative for a positive modulus), i.e. the reference's `((a % m) + m) % m`.
pub fn modulus(a: &BigInt, m: &BigInt) -> BigInt {
    let r = a % m; // truncated: sign of a
    if !r.is_zero() && (r.sign() == Sign::Minus) != (m.sign() == Sign::Minus) {
        r + m
    } else {
        r
    }
}

/// `a^-1 mod q` in `[0, q)`; `Err` for `a == 0`.  Like the reference it does not test `gcd(a, q) == 1`: for a
/// non-invertible `a` the result is the Bezout coefficient the Euclidean remainder sequence ends with.
pub fn modinv(a: &BigInt, q: &BigInt) -> Result<BigInt, String> {
    if a.is_zero() {
        return Err("no mod inv of Zero".to_string());
    }
    // remainders r0 > r1 > ... of (q, a) and the coefficients t_k with r_k == t_k * a (mod q)
    let (mut r0, mut r1) = (q.clone(), a.clone());
    let (mut t0, mut t1) = (BigInt::zero(), BigInt::one());
    while !r1.is_zero() {
        let quot = &r0 / &r1;
        let t2 = &t0 - &quot * &t1;
        let r2 = modulus(&r0, &r1);
        r0 = r1;
        r1 = r2;
        t0 = t1;
        t1 = t2;
    }
    if t0.sign() == Sign::Minus {
        t0 = modulus(&t0, q);
    }
    Ok(t0)
}

pub fn concatenate_arrays<T: Clone>(x: &[T], y: &[T]) -> Vec<T> {
    let mut v = Vec::with_capacity(x.len() + y.len());
    v.extend_from_slice(x);
    v.extend_from_slice(y);
    v
}

/// Euler's criterion: -1 when `a` is a quadratic non-residue mod the odd prime `q`, 1 otherwise (also for `a = 0`).
pub fn legendre_symbol(a: &BigInt, q: &BigInt) -> i32 {
    let minus_one = q - BigInt::one();
    if a.modpow(&(&minus_one >> 1), q) == minus_one {
        -1
    } else {
        1
    }
}

/// Tonelli-Shanks.  q - 1 = s * 2^e with s odd, z = n^s for the least non-residue n >= 2 generates the 2-Sylow subgroup;
/// starting from x = a^((s+1)/2), b = a^s (so x^2 = a b) the order 2^m of b is brought down to 1 by multiplying x with
/// z^(2^(r-m-1)) and b with its square, r being the order exponent of the current z.
fn tonelli_shanks(a: &BigInt, q: &BigInt) -> Result<BigInt, String> {
    let one = BigInt::one();
    let two = BigInt::from(2);
    if legendre_symbol(a, q) != 1 || a.is_zero() || *q == two {
        return Err("not a mod p square".to_string());
    }
    if q % BigInt::from(4) == BigInt::from(3) {
        return Ok(a.modpow(&((q + &one) >> 2), q));
    }
    let mut s = q - &one;
    let mut r: u64 = 0;
    while (&s % &two).is_zero() {
        s >>= 1;
        r += 1;
    }
    let mut n = two.clone();
    while legendre_symbol(&n, q) != -1 {
        n += &one;
    }
    let mut x = a.modpow(&((&s + &one) >> 1), q);
    let mut b = a.modpow(&s, q);
    let mut z = n.modpow(&s, q);
    if b.is_zero() {
        // a is a non-zero multiple of q: the reference's order search never terminates here; 0 has no root by its rule
        return Err("not a mod p square".to_string());
    }
    loop {
        // order of b in the 2-Sylow subgroup: least m with b^(2^m) == 1
        let mut m: u64 = 0;
        let mut t = b.clone();
        while t != one {
            t = (&t * &t) % q;
            m += 1;
        }
        if m == 0 {
            return Ok(x);
        }
        let mut w = z; // z^(2^(r-m-1))
        for _ in 0..(r - m - 1) {
            w = (&w * &w) % q;
        }
        z = (&w * &w) % q;
        x = (x * &w) % q;
        b = (b * &z) % q;
        r = m;
    }
}

pub fn modsqrt(a: &BigInt, q: &BigInt) -> Result<BigInt, String> {
    tonelli_shanks(a, q)
}

/// The reference keeps a second formulation of the same algorithm under this name (`src/utils.rs:164-213`); both start
/// from the least non-residue and walk the same sequence of corrections, so they return the same root.
#[allow(dead_code)]
pub fn modsqrt_v2(a: &BigInt, q: &BigInt) -> Result<BigInt, String> {
    tonelli_shanks(a, q)
}
