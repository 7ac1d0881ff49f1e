// DEBUG HARNESS (tests only): runs the product's __host__ __device__ field code
// on the CPU with limb/value-bound assertions enabled.  Not part of libbjj_hip.so.
#define BJJ_DEBUG_BOUNDS 1
#include <string.h>
#include "../../babyjubjub-rs_amd/csrc/fr.hpp"
using namespace bjj;
static Fr load(const uint8_t* b) { u32 w[8]; memcpy(w, b, 32); return fr_to_mont_words(w); }
static void store(uint8_t* b, const Fr& a) { u32 w[8]; fr_from_mont_words(a, w); memcpy(b, w, 32); }
extern "C" {
void emul_fr_mul(const uint8_t* a, const uint8_t* b, uint8_t* o) { store(o, fr_mul(load(a), load(b))); }
void emul_fr_sqr(const uint8_t* a, uint8_t* o) { store(o, fr_sqr(load(a))); }
void emul_fr_add(const uint8_t* a, const uint8_t* b, uint8_t* o) { store(o, fr_add(load(a), load(b))); }
void emul_fr_sub(const uint8_t* a, const uint8_t* b, uint8_t* o) { store(o, fr_sub(load(a), load(b))); }
void emul_fr_sub8(const uint8_t* a, const uint8_t* b, uint8_t* o) { store(o, fr_sub8(load(a), load(b))); }
void emul_fr_inv(const uint8_t* a, uint8_t* o) { store(o, fr_inv_fermat(load(a))); }
int emul_fr_eq(const uint8_t* a, const uint8_t* b) { return fr_eq(load(a), load(b)); }
// (a*b - c) * (a + b) + lazy chains, exercising value growth
void emul_fr_chain(const uint8_t* a, const uint8_t* b, const uint8_t* c, uint8_t* o) {
  Fr x = load(a), y = load(b), z = load(c);
  Fr s = fr_add(x, y);            // < 4r
  Fr d = fr_sub(fr_mul(x, y), z); // < 6r
  Fr e = fr_sub8(d, s);           // < 14r ... too big for mul with >1r? e*x: 14r*2r ok
  store(o, fr_mul(e, x));
}
// plain (non-Montgomery) round trip of an arbitrary 256-bit integer through the limb converters
void emul_words_roundtrip(const uint8_t* a, uint8_t* o) { u32 w[8]; memcpy(w, a, 32); Fr f = fr_from_words(w); u32 v[8]; fr_to_words(f, v); memcpy(o, v, 32); }
}
extern "C" void emul_fr_inv_gcd(const uint8_t* a, uint8_t* o) { store(o, fr_inv_gcd(load(a))); }
// raw (no Montgomery conversion of the input): exercises arbitrary N-form inputs < 16r
extern "C" void emul_fr_inv_gcd_lazy(const uint8_t* a, const uint8_t* b, uint8_t* o) { store(o, fr_inv_gcd(fr_add(fr_sub8(load(a), load(b)), load(b)))); }
