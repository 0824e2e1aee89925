/* TEST INFRASTRUCTURE -- CPU oracle, not product code.  Only tests/, the
 * smoke() check and bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * PARITY UNPINNED at the byte level: the reference's arithmetic for this path
 * lives in risc0-zkp 1.0.1 / risc0-core 1.0.1 (reference Cargo.lock:7243,
 * :7171), which are not vendored under /root/reference and cannot be built here
 * (no Rust toolchain).  This file restates the published algorithm:
 *   - BabyBear, p = 15*2^27+1, Montgomery form R = 2^32 (risc0-core
 *     field/baby_bear.rs), call sites in the reference:
 *     provers/risc0/driver/src/bonsai.rs:246-271.
 *   - quartic extension Fp[x]/(x^4 - W), W = -11 (x^4+11) by default.
 * It is pinned by exact big-integer arithmetic in tests/test_oracle_field.py.
 */
#ifndef OR_FIELD_H
#define OR_FIELD_H
#include <stdint.h>
#include <stddef.h>

#define OR_P 2013265921u
#define OR_M 0x88000001u      /* p^-1 mod 2^32 */
#define OR_R2 1172168163u     /* 2^64 mod p */
#define OR_INVALID 0xffffffffu
#define OR_BETA 11u
/* Montgomery form of W in Fp[x]/(x^4 - W): p - 11 unless or_set_params says otherwise (or_params.c) */
extern uint32_t g_or_wm;

typedef uint32_t fp; /* Montgomery residue, canonical range [0,p) */
typedef struct { fp c[4]; } fp4;

static inline fp fp_add(fp a, fp b) { uint32_t r = a + b; return r >= OR_P ? r - OR_P : r; }
static inline fp fp_sub(fp a, fp b) { uint32_t r = a - b; return a < b ? r + OR_P : r; }
static inline fp fp_neg(fp a) { return a ? OR_P - a : 0; }
static inline fp fp_mul(fp a, fp b) {
    uint64_t o = (uint64_t)a * b;
    uint32_t low = 0u - (uint32_t)o;
    uint32_t red = OR_M * low;
    o += (uint64_t)red * OR_P;
    uint32_t r = (uint32_t)(o >> 32);
    return r >= OR_P ? r - OR_P : r;
}
static inline fp fp_from_u32(uint32_t x) { return fp_mul(x % OR_P, OR_R2); }
static inline uint32_t fp_to_u32(fp a) { return fp_mul(a, 1u); }
static inline fp fp_pow(fp a, uint64_t e) {
    fp r = fp_from_u32(1);
    while (e) { if (e & 1) r = fp_mul(r, a); a = fp_mul(a, a); e >>= 1; }
    return r;
}
static inline fp fp_inv(fp a) { return fp_pow(a, OR_P - 2); }

static inline fp4 fp4_zero(void) { fp4 r = {{0, 0, 0, 0}}; return r; }
static inline fp4 fp4_from_fp(fp a) { fp4 r = {{a, 0, 0, 0}}; return r; }
static inline fp4 fp4_one(void) { return fp4_from_fp(fp_from_u32(1)); }
static inline fp4 fp4_add(fp4 a, fp4 b) { fp4 r; for (int i = 0; i < 4; i++) r.c[i] = fp_add(a.c[i], b.c[i]); return r; }
static inline fp4 fp4_sub(fp4 a, fp4 b) { fp4 r; for (int i = 0; i < 4; i++) r.c[i] = fp_sub(a.c[i], b.c[i]); return r; }
static inline fp4 fp4_scale(fp4 a, fp s) { fp4 r; for (int i = 0; i < 4; i++) r.c[i] = fp_mul(a.c[i], s); return r; }
static inline int fp4_eq(fp4 a, fp4 b) { return a.c[0]==b.c[0] && a.c[1]==b.c[1] && a.c[2]==b.c[2] && a.c[3]==b.c[3]; }
/* schoolbook product reduced by x^4 = -11 */
static inline fp4 fp4_mul(fp4 a, fp4 b) {
    fp nbeta = g_or_wm;
    fp4 r;
    r.c[0] = fp_add(fp_mul(a.c[0], b.c[0]),
                    fp_mul(nbeta, fp_add(fp_add(fp_mul(a.c[1], b.c[3]), fp_mul(a.c[2], b.c[2])), fp_mul(a.c[3], b.c[1]))));
    r.c[1] = fp_add(fp_add(fp_mul(a.c[0], b.c[1]), fp_mul(a.c[1], b.c[0])),
                    fp_mul(nbeta, fp_add(fp_mul(a.c[2], b.c[3]), fp_mul(a.c[3], b.c[2]))));
    r.c[2] = fp_add(fp_add(fp_add(fp_mul(a.c[0], b.c[2]), fp_mul(a.c[1], b.c[1])), fp_mul(a.c[2], b.c[0])),
                    fp_mul(nbeta, fp_mul(a.c[3], b.c[3])));
    r.c[3] = fp_add(fp_add(fp_mul(a.c[0], b.c[3]), fp_mul(a.c[1], b.c[2])),
                    fp_add(fp_mul(a.c[2], b.c[1]), fp_mul(a.c[3], b.c[0])));
    return r;
}
static inline fp4 fp4_pow(fp4 a, uint64_t e) {
    fp4 r = fp4_one();
    while (e) { if (e & 1) r = fp4_mul(r, a); a = fp4_mul(a, a); e >>= 1; }
    return r;
}
/* a^-1 = a^(p^4-2); exponent handled as four base-p digits via Frobenius-free
 * square-and-multiply on a 124-bit exponent split in two 64-bit halves. */
static inline fp4 fp4_inv(fp4 a) {
    /* p^4 - 2 as 128-bit: compute with unsigned __int128 */
    unsigned __int128 e = (unsigned __int128)OR_P * OR_P;
    e = e * OR_P * OR_P - 2;
    fp4 r = fp4_one();
    while (e) { if (e & 1) r = fp4_mul(r, a); a = fp4_mul(a, a); e >>= 1; }
    return r;
}
#endif
