//! Known-answer vectors from Plonky3 (rev 88ea2b86, the revision raiko's Cargo.lock pins through sp1) for the path
//! `client.prove(&pk, stdin)` of provers/sp1/driver/src/lib.rs:44-57 runs through: the width-16 BabyBear Poseidon2, the
//! sponge and compression built on it, the coset LDE and the FRI fold of the two-adic PCS, and one whole uni-stark proof
//! of Plonky3's own Fibonacci AIR.  Everything Plonky3-internal below is written from recollection of that revision
//! (marked RECALLED): `cargo check` first and adjust paths / signatures; the FILE FORMAT is what must not change --
//! tests/golden/p3_vector_format.py reads exactly what `Out` writes here.
//!
//! Every file: little-endian u32 words, word 0 = 0x31564B52 ("RKV1"), word 1 = kind.  Field elements are written as
//! MONTGOMERY words with R = 2^32 (`mont(x) = x * 2^32 mod p` of the canonical value): the representation of every
//! buffer of include/raiko_hip.h.
//!   kind 5  p3_poseidon2.bin  width rp | rc_ext[8 * width] rc_int[rp] diag[width] (the instance's constants as the
//!                             matrices x -> x + rc and 1 1^T + diag(d) need them, canonical values in Montgomery form) |
//!                             3 x { in[width] out[width] } of the permutation | row_len row[row_len] digest[8]
//!                             (PaddingFreeSponge<_, 16, 8, 8>) | left[8] right[8] out[8] (TruncatedPermutation<_, 2, 8, 16>)
//!   kind 6  p3_pcs.bin        log_h w trace[h * w] (row-major) | lde[2h * w] = coset_lde_batch(trace, 1, generator)
//!                             .bit_reverse_rows() | n beta[4] evals[4 n] folded[4 (n / 2)] (fold_even_odd on
//!                             bit-reversed evaluations of an extension-valued vector)
//!   kind 7  p3_fib_proof.bin  log_n queries pow_bits log_blowup | public[3] | trace[2 n] (row-major) |
//!                             proof_words proof[] in the layout of rk_p3_prove (include/raiko_hip.h):
//!                             1 | log_n | trace root 8 | quotient root 8 | trace_local 4 x 2 | trace_next 4 x 2 |
//!                             chunk 4 x 4 | n_rounds | roots 8 each | final_poly 4 | pow witness (canonical) |
//!                             per query: trace row 2, path 8 per level; quotient row 4, path; per round sibling 4, path
use std::{fs, path::Path};

use p3_air::{Air, AirBuilder, AirBuilderWithPublicValues, BaseAir};
use p3_baby_bear::{BabyBear, DiffusionMatrixBabyBear};
use p3_challenger::DuplexChallenger;
use p3_commit::ExtensionMmcs;
use p3_dft::{Radix2DitParallel, TwoAdicSubgroupDft};
use p3_field::{extension::BinomialExtensionField, AbstractField, Field, PrimeField32};
use p3_fri::{FriConfig, TwoAdicFriPcs};
use p3_matrix::{bitrev::BitReversableMatrix, dense::RowMajorMatrix, Matrix};
use p3_merkle_tree::FieldMerkleTreeMmcs;
use p3_poseidon2::{Poseidon2, Poseidon2ExternalMatrixGeneral};
use p3_symmetric::{CryptographicHasher, PaddingFreeSponge, Permutation, PseudoCompressionFunction, TruncatedPermutation};
use p3_uni_stark::{prove, verify, StarkConfig};
use rand::{Rng, SeedableRng};
use rand_chacha::ChaCha20Rng;

type Val = BabyBear;
type Challenge = BinomialExtensionField<Val, 4>;
type Perm = Poseidon2<Val, Poseidon2ExternalMatrixGeneral, DiffusionMatrixBabyBear, 16, 7>;
type MyHash = PaddingFreeSponge<Perm, 16, 8, 8>;
type MyCompress = TruncatedPermutation<Perm, 2, 8, 16>;
type ValMmcs = FieldMerkleTreeMmcs<<Val as Field>::Packing, <Val as Field>::Packing, MyHash, MyCompress, 8>;
type ChallengeMmcs = ExtensionMmcs<Val, Challenge, ValMmcs>;
type Dft = Radix2DitParallel;
type Challenger = DuplexChallenger<Val, Perm, 16, 8>;
type Pcs = TwoAdicFriPcs<Val, Dft, ValMmcs, ChallengeMmcs>;
type MyConfig = StarkConfig<Pcs, Challenge, Challenger>;

const MAGIC: u32 = 0x3156_4B52;
const P: u64 = 15 * (1 << 27) + 1;

fn mont(v: Val) -> u32 {
    (((v.as_canonical_u32() as u64) << 32) % P) as u32
}

struct Out(Vec<u32>);
impl Out {
    fn new(kind: u32) -> Self {
        Out(vec![MAGIC, kind])
    }
    fn word(&mut self, w: u32) {
        self.0.push(w);
    }
    fn elems(&mut self, e: &[Val]) {
        self.0.extend(e.iter().map(|v| mont(*v)));
    }
    fn ext(&mut self, e: &Challenge) {
        // RECALLED: BinomialExtensionField exposes its base coefficients through AbstractExtensionField::as_base_slice
        use p3_field::AbstractExtensionField;
        self.elems(<Challenge as AbstractExtensionField<Val>>::as_base_slice(e));
    }
    fn save(&self, dir: &Path, name: &str) -> anyhow::Result<()> {
        let bytes: Vec<u8> = self.0.iter().flat_map(|w| w.to_le_bytes()).collect();
        fs::write(dir.join(name), bytes)?;
        println!("{name}: {} words", self.0.len());
        Ok(())
    }
}

/// The instance sp1 uses comes from sp1-primitives (RC_16_30 + the BabyBear internal diagonal); any instance pins the
/// arithmetic as long as its constants travel with the vectors, so this program draws them from a fixed ChaCha stream and
/// writes them into the file (the tests configure oracle and library with exactly these).
fn instance(rng: &mut ChaCha20Rng) -> (Perm, Vec<[Val; 16]>, Vec<Val>) {
    let rc_ext: Vec<[Val; 16]> = (0..8).map(|_| core::array::from_fn(|_| Val::from_canonical_u32(rng.gen_range(0..P as u32)))).collect();
    let rc_int: Vec<Val> = (0..13).map(|_| Val::from_canonical_u32(rng.gen_range(0..P as u32))).collect();
    // RECALLED: Poseidon2::new(rounds_f, external_constants, external_layer, rounds_p, internal_constants, internal_layer)
    let perm = Perm::new(8, rc_ext.clone(), Poseidon2ExternalMatrixGeneral, 13, rc_int.clone(), DiffusionMatrixBabyBear::default());
    (perm, rc_ext, rc_int)
}

/// 1 1^T + diag(d): column i of the internal layer's matrix minus the all-ones part, read off the implementation itself
/// (so whatever convention DiffusionMatrixBabyBear uses internally, the file states the matrix it realises)
fn internal_diag(perm_layer: &DiffusionMatrixBabyBear) -> [Val; 16] {
    core::array::from_fn(|i| {
        let mut e = [Val::zero(); 16];
        e[i] = Val::one();
        // RECALLED: the internal layer is a Permutation<[Val; 16]> applying state -> (1 1^T + diag) state
        perm_layer.permute_mut(&mut e);
        e[i] - Val::one()
    })
}

fn poseidon2_file(dir: &Path) -> anyhow::Result<Perm> {
    let mut rng = ChaCha20Rng::seed_from_u64(0x7033);
    let (perm, rc_ext, rc_int) = instance(&mut rng);
    let mut o = Out::new(5);
    o.word(16);
    o.word(13);
    for r in &rc_ext {
        o.elems(r);
    }
    o.elems(&rc_int);
    o.elems(&internal_diag(&DiffusionMatrixBabyBear::default()));
    for _ in 0..3 {
        let x: [Val; 16] = core::array::from_fn(|_| Val::from_canonical_u32(rng.gen_range(0..P as u32)));
        o.elems(&x);
        o.elems(&perm.permute(x));
    }
    let row: Vec<Val> = (0..37).map(|_| Val::from_canonical_u32(rng.gen_range(0..P as u32))).collect();
    o.word(row.len() as u32);
    o.elems(&row);
    o.elems(&MyHash::new(perm.clone()).hash_iter(row.iter().copied()));
    let l: [Val; 8] = core::array::from_fn(|_| Val::from_canonical_u32(rng.gen_range(0..P as u32)));
    let r: [Val; 8] = core::array::from_fn(|_| Val::from_canonical_u32(rng.gen_range(0..P as u32)));
    o.elems(&l);
    o.elems(&r);
    o.elems(&MyCompress::new(perm.clone()).compress([l, r]));
    o.save(dir, "p3_poseidon2.bin")?;
    Ok(perm)
}

fn pcs_file(dir: &Path) -> anyhow::Result<()> {
    let mut rng = ChaCha20Rng::seed_from_u64(0x7034);
    let (log_h, w) = (6usize, 5usize);
    let h = 1 << log_h;
    let trace = RowMajorMatrix::new((0..h * w).map(|_| Val::from_canonical_u32(rng.gen_range(0..P as u32))).collect(), w);
    let mut o = Out::new(6);
    o.word(log_h as u32);
    o.word(w as u32);
    o.elems(&trace.values);
    // what TwoAdicFriPcs::commit does per matrix: coset LDE by the blow-up with shift = Val::generator(), rows bit-reversed
    let lde = Radix2DitParallel.coset_lde_batch(trace, 1, Val::generator()).bit_reverse_rows().to_row_major_matrix();
    o.elems(&lde.values);
    // p3-fri fold_even_odd on an extension-valued vector in bit-reversed order
    let n = 32usize;
    let beta = Challenge::from_base_fn(|_| Val::from_canonical_u32(rng.gen_range(0..P as u32)));
    let evals: Vec<Challenge> = (0..n).map(|_| Challenge::from_base_fn(|_| Val::from_canonical_u32(rng.gen_range(0..P as u32)))).collect();
    o.word(n as u32);
    o.ext(&beta);
    for e in &evals {
        o.ext(e);
    }
    for e in p3_fri::fold_even_odd(evals, beta) {
        o.ext(&e);
    }
    o.save(dir, "p3_pcs.bin")
}

/// Plonky3's uni-stark test AIR (uni-stark/tests/fib_air.rs)
struct FibonacciAir;
impl<F> BaseAir<F> for FibonacciAir {
    fn width(&self) -> usize {
        2
    }
}
impl<AB: AirBuilderWithPublicValues> Air<AB> for FibonacciAir {
    fn eval(&self, builder: &mut AB) {
        let main = builder.main();
        let pis = builder.public_values();
        let (a, b, x) = (pis[0], pis[1], pis[2]);
        let (local, next) = (main.row_slice(0), main.row_slice(1));
        let mut first = builder.when_first_row();
        first.assert_eq(local[0], a);
        first.assert_eq(local[1], b);
        let mut tr = builder.when_transition();
        tr.assert_eq(local[1], next[0]);
        tr.assert_eq(local[0] + local[1], next[1]);
        builder.when_last_row().assert_eq(local[1], x);
    }
}

fn fib_file(dir: &Path, perm: Perm) -> anyhow::Result<()> {
    let (log_n, queries, pow_bits, log_blowup) = (6usize, 8usize, 6usize, 1usize);
    let n = 1 << log_n;
    let mut rows = vec![Val::zero(); 2 * n];
    let (mut l, mut r) = (Val::zero(), Val::one());
    for i in 0..n {
        rows[2 * i] = l;
        rows[2 * i + 1] = r;
        (l, r) = (r, l + r);
    }
    let public = vec![Val::zero(), Val::one(), rows[2 * n - 1]];
    let trace = RowMajorMatrix::new(rows.clone(), 2);
    let val_mmcs = ValMmcs::new(MyHash::new(perm.clone()), MyCompress::new(perm.clone()));
    let fri = FriConfig { log_blowup, num_queries: queries, proof_of_work_bits: pow_bits, mmcs: ChallengeMmcs::new(val_mmcs.clone()) };
    let pcs = Pcs::new(log_n, Radix2DitParallel, val_mmcs, fri);
    let config = MyConfig::new(pcs);
    let mut ch = Challenger::new(perm.clone());
    let proof = prove(&config, &FibonacciAir, &mut ch, trace, &public);
    verify(&config, &FibonacciAir, &mut Challenger::new(perm), &proof, &public).expect("plonky3 refuses its own proof");

    let mut o = Out::new(7);
    for v in [log_n, queries, pow_bits, log_blowup] {
        o.word(v as u32);
    }
    o.elems(&public);
    o.elems(&rows);
    // ---- the proof in rk_p3_prove's word order (RECALLED field names of p3-uni-stark Proof / p3-fri FriProof)
    let mut w = Out(vec![]);
    w.word(1);
    w.word(proof.degree_bits as u32);
    let digest = |out: &mut Out, h: &p3_symmetric::Hash<Val, Val, 8>| {
        let a: [Val; 8] = (*h).into();
        out.elems(&a);
    };
    digest(&mut w, &proof.commitments.trace);
    digest(&mut w, &proof.commitments.quotient_chunks);
    for e in &proof.opened_values.trace_local {
        w.ext(e);
    }
    for e in &proof.opened_values.trace_next {
        w.ext(e);
    }
    for chunk in &proof.opened_values.quotient_chunks {
        for e in chunk {
            w.ext(e);
        }
    }
    let fp = &proof.opening_proof;
    w.word(fp.commit_phase_commits.len() as u32);
    for c in &fp.commit_phase_commits {
        digest(&mut w, c);
    }
    w.ext(&fp.final_poly);
    w.word(fp.pow_witness.as_canonical_u32());
    for q in &fp.query_proofs {
        // input_proof: one BatchOpening per round of the PCS (traces, then quotient chunks): opened rows per matrix, then the path
        for batch in &q.input_proof {
            for row in &batch.opened_values {
                w.elems(row);
            }
            for sib in &batch.opening_proof {
                w.elems(sib);
            }
        }
        for step in &q.commit_phase_openings {
            w.ext(&step.sibling_value);
            for sib in &step.opening_proof {
                w.elems(sib);
            }
        }
    }
    o.word(w.0.len() as u32);
    o.0.extend_from_slice(&w.0);
    o.save(dir, "p3_fib_proof.bin")
}

fn main() -> anyhow::Result<()> {
    let dir = std::env::args().nth(1).unwrap_or_else(|| ".".into());
    let dir = Path::new(&dir);
    fs::create_dir_all(dir)?;
    let perm = poseidon2_file(dir)?;
    pcs_file(dir)?;
    fib_file(dir, perm)
}
