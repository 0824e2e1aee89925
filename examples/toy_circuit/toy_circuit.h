/* toy_circuit.h -- a complete small circuit behind rk_circuit_hooks (include/raiko_hip.h): how a
 * host binds CircuitHal::accumulate / eval_check and CircuitDef::poly_ext to libraiko_hip.so.
 * It stands where risc0-circuit-rv32im 1.0.1 stands behind `session.prove()` (reference
 * provers/risc0/driver/src/bonsai.rs:271); that crate is not vendored in the reference, so this is
 * NOT the rv32im circuit -- it is the same interface on a circuit small enough to read:
 *
 *   code  c0/c1/c2 = selectors of row 0 / row 1 / the last row          (Wc >= 3, rest free)
 *   data  d0 Fibonacci, d1 = d0 * d0[-1], d3 a permutation of d2        (Wd >= 4, rest free)
 *   accum A = (a0..a3) running product of (m + d2)/(m + d3), m = (mix0..mix3);
 *         a_k = mix[k % n_mix] * d[k % Wd] for 4 <= k < Wa               (Wa >= 4, n_mix >= 4)
 *   K0 = (1 - c0 - c1)(d0 - d0[-1] - d0[-2])      K1 = d1 - d0 d0[-1]
 *   K2 = A (m + d3) - ((1 - c0) A[-1] + c0)(m + d2)    K3 = c2 (A - 1)    K_k = a_k - mix_k d_k
 *   check(x) = sum_k poly_mix^k K_k(x) / (x^N - 1)
 */
#ifndef TOY_CIRCUIT_H
#define TOY_CIRCUIT_H
#include "raiko_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* hooks for rk_segment.hooks (device kernels on the prover's stream; `user` unused) */
const rk_circuit_hooks* toy_circuit_hooks(void);
/* rk_verify_opts.poly_ext: the mixed constraint polynomial on the tap openings (host code) */
int toy_circuit_poly_ext(void* user, const rk_segment* pub, const uint32_t poly_mix[4], const uint32_t* eval_u_ext,
                         size_t n_taps, const uint32_t* mix, uint32_t n_mix, uint32_t out_ext[4]);
#ifdef __cplusplus
}
#endif
#endif
