//! Known-answer vectors from risc0-zkp / risc0-core 1.0.1 for the hot path raiko reaches through
//! `session.prove()` (provers/risc0/driver/src/bonsai.rs:271).  Everything risc0-internal below is written from
//! recollection of the 1.0.1 crates (marked RECALLED): `cargo check` first and adjust paths / signatures; the FILE
//! FORMAT is what must not change -- tests/golden/risc0_vector_format.py reads exactly what `Out` writes here.
//!
//! Every file: little-endian u32 words, word 0 = 0x31564B52 ("RKV1"), word 1 = kind.  Field elements are risc0's own
//! in-memory words (`Elem` is a Montgomery residue, R = 2^32), which is the representation of every buffer of
//! include/raiko_hip.h.
//!   kind 1  poseidon2.bin  3 x { in[24] out[24] } of poseidon2_mix | a[8] b[8] hash_pair[8] |
//!                          rows cols matrix[cols * rows] (column-major) digests[rows * 8] of hash_rows
//!   kind 2  ntt.bin        k count evals[count << k] | coeffs[count << k] (batch_interpolate_ntt) |
//!                          shifted[count << k] (zk_shift) | expanded[count << (k + 2)] (batch_expand_into_evaluate_ntt)
//!   kind 3  rng.bin        d1[8] d2[8] | after mix(d1): random_bits(20) x 4, random_elem x 4 |
//!                          after mix(d2): random_ext[4], random_bits(10)
//!   kind 4  seal.bin       po2 n_globals n_accum_mix group_size[3] n_regs n_combos n_backs |
//!                          reg_group[] reg_offset[] reg_combo[] combo_off[n_combos + 1] combo_backs[] |
//!                          proof_system_info[4] circuit_info[4] (16 bytes each, LE words) | globals[] |
//!                          accum[N * w0] code[N * w1] data[N * w2] (column-major) | check[4 * 4N] |
//!                          seal_words seal[]
use std::{fs, path::Path};

use risc0_core::field::{
    baby_bear::{BabyBear, BabyBearElem as Elem, BabyBearExtElem as ExtElem},
    Elem as _, ExtElem as _,
};
use risc0_zkp::{
    adapter::{CircuitInfo, TapsProvider},
    core::{
        digest::Digest,
        hash::poseidon2::{poseidon2_mix, Poseidon2HashSuite, CELLS},
    },
    hal::{cpu::CpuHal, Buffer, CircuitHal, Hal},
    prove::Prover,
    taps::{TapData, TapSet},
};

const MAGIC: u32 = 0x3156_4B52;
const P: u64 = 15 * (1 << 27) + 1;

struct Out(Vec<u32>);
impl Out {
    fn new(kind: u32) -> Self {
        Out(vec![MAGIC, kind])
    }
    fn word(&mut self, w: u32) {
        self.0.push(w);
    }
    fn elems(&mut self, e: &[Elem]) {
        self.0.extend_from_slice(bytemuck::cast_slice::<Elem, u32>(e));
    }
    fn words(&mut self, w: &[u32]) {
        self.0.extend_from_slice(w);
    }
    fn save(&self, dir: &Path, name: &str) -> anyhow::Result<()> {
        fs::write(dir.join(name), bytemuck::cast_slice::<u32, u8>(&self.0))?;
        println!("{name}: {} words", self.0.len());
        Ok(())
    }
}

/// deterministic, easy to restate anywhere: canonical value ((2654435761 * (i + 1) + salt) mod 2^32) mod p
fn elem(i: usize, salt: u32) -> Elem {
    let v = (2654435761u64.wrapping_mul(i as u64 + 1).wrapping_add(salt as u64)) & 0xffff_ffff;
    Elem::new((v % P) as u32)
}
fn elems(n: usize, salt: u32) -> Vec<Elem> {
    (0..n).map(|i| elem(i, salt)).collect()
}
fn digest_words(d: &Digest) -> [u32; 8] {
    let mut w = [0u32; 8];
    w.copy_from_slice(d.as_words());
    w
}

fn main() -> anyhow::Result<()> {
    let dir = std::env::args().nth(1).unwrap_or_else(|| ".".to_owned());
    let dir = Path::new(&dir);
    fs::create_dir_all(dir)?;
    let suite = Poseidon2HashSuite::new_suite();
    let hal = CpuHal::<BabyBear>::new(suite.clone());
    let hashfn = &suite.hashfn;

    // ---- kind 1: the permutation, the 2-to-1 compression, the row sponge
    let mut o = Out::new(1);
    for k in 0..3u32 {
        let mut cells: [Elem; CELLS] = core::array::from_fn(|i| elem(i, 1000 * (k + 1)));
        o.elems(&cells);
        poseidon2_mix(&mut cells);
        o.elems(&cells);
    }
    let a = hashfn.hash_elem_slice(&elems(5, 7));
    let b = hashfn.hash_elem_slice(&elems(9, 8));
    o.words(&digest_words(&a));
    o.words(&digest_words(&b));
    o.words(&digest_words(&hashfn.hash_pair(&a, &b)));
    let (rows, cols) = (8usize, 40usize);
    let matrix = hal.copy_from_elem("matrix", &elems(rows * cols, 9));
    let digests = hal.alloc_digest("digests", rows);
    hal.hash_rows(&digests, &matrix);
    o.word(rows as u32);
    o.word(cols as u32);
    matrix.view(|m| o.elems(m));
    digests.view(|d| d.iter().for_each(|x| o.words(&digest_words(x))));
    o.save(dir, "poseidon2.bin")?;

    // ---- kind 2: interpolate, zk shift, 4x expansion of three columns of 2^10 evaluations
    let (k, count) = (10usize, 3usize);
    let n = 1usize << k;
    let mut o = Out::new(2);
    o.word(k as u32);
    o.word(count as u32);
    let io = hal.copy_from_elem("io", &elems(count * n, 21));
    io.view(|v| o.elems(v));
    hal.batch_interpolate_ntt(&io, count);
    io.view(|v| o.elems(v));
    hal.zk_shift(&io, count);
    io.view(|v| o.elems(v));
    let wide = hal.alloc_elem("wide", count * n * 4);
    hal.batch_expand_into_evaluate_ntt(&wide, &io, count, 2);
    wide.view(|v| o.elems(v));
    o.save(dir, "ntt.bin")?;

    // ---- kind 3: the Fiat-Shamir generator
    let mut o = Out::new(3);
    let d1 = hashfn.hash_elem_slice(&elems(3, 31));
    let d2 = hashfn.hash_elem_slice(&elems(4, 32));
    o.words(&digest_words(&d1));
    o.words(&digest_words(&d2));
    let mut rng = suite.rng.new_rng();
    rng.mix(&d1);
    for _ in 0..4 {
        o.word(rng.random_bits(20));
    }
    for _ in 0..4 {
        o.elems(&[rng.random_elem()]);
    }
    rng.mix(&d2);
    let e: ExtElem = rng.random_ext_elem();
    o.elems(e.subelems());
    o.word(rng.random_bits(10));
    o.save(dir, "rng.bin")?;

    // ---- kind 4: one whole seal of a synthetic circuit of 2^10 cycles (risc0-circuit-rv32im prove_segment's
    // call sequence on the zkp Prover, with accum and the check polynomial given instead of computed)
    seal_vector(dir, &hal, &suite)?;
    Ok(())
}

/// The circuit side of `Prover::finalize`, with the check evaluations handed in.
struct Given {
    check: Vec<Elem>,
}
impl CircuitHal<CpuHal<BabyBear>> for Given {
    // RECALLED signature (risc0-zkp 1.0.1 hal/mod.rs)
    fn eval_check(
        &self,
        check: &<CpuHal<BabyBear> as Hal>::Buffer<Elem>,
        _groups: &[&<CpuHal<BabyBear> as Hal>::Buffer<Elem>],
        _globals: &[&<CpuHal<BabyBear> as Hal>::Buffer<Elem>],
        _poly_mix: ExtElem,
        _po2: usize,
        _steps: usize,
    ) {
        check.view_mut(|c| c.copy_from_slice(&self.check));
    }
    fn accumulate(
        &self,
        _ctrl: &<CpuHal<BabyBear> as Hal>::Buffer<Elem>,
        _io: &<CpuHal<BabyBear> as Hal>::Buffer<Elem>,
        _data: &<CpuHal<BabyBear> as Hal>::Buffer<Elem>,
        _mix: &<CpuHal<BabyBear> as Hal>::Buffer<Elem>,
        _accum: &<CpuHal<BabyBear> as Hal>::Buffer<Elem>,
        _steps: usize,
    ) {
    }
}

fn seal_vector(dir: &Path, hal: &CpuHal<BabyBear>, suite: &risc0_zkp::core::hash::HashSuite<BabyBear>) -> anyhow::Result<()> {
    let po2 = 10usize;
    let n = 1usize << po2;
    let group_size = [4usize, 4, 8]; // accum, code, data
    let (n_globals, n_accum_mix) = (6usize, 5usize);
    // combos: 0 = {0}, 1 = {0, 1}; accum registers and data registers 0, 1 read one row back as well
    let combo_of = |g: usize, o: usize| -> usize { usize::from(g == 0 || (g == 2 && o < 2)) };
    let combo_backs: [&[u16]; 2] = [&[0], &[0, 1]];
    let mut taps = Vec::new();
    let (mut reg_group, mut reg_offset, mut reg_combo) = (vec![], vec![], vec![]);
    let mut group_begin = [0usize; 4];
    for g in 0..3 {
        group_begin[g] = taps.len();
        for o in 0..group_size[g] {
            let c = combo_of(g, o);
            reg_group.push(g as u32);
            reg_offset.push(o as u32);
            reg_combo.push(c as u32);
            for (j, &back) in combo_backs[c].iter().enumerate() {
                // RECALLED: TapData { offset, back, group, combo, skip }, skip = taps of this register (on its first tap)
                taps.push(TapData { offset: o as u16, back, group: g, combo: c as u8, skip: if j == 0 { combo_backs[c].len() as u8 } else { 0 } });
            }
        }
    }
    group_begin[3] = taps.len();
    let combo_taps: Vec<u16> = combo_backs.iter().flat_map(|b| b.iter().copied()).collect();
    let combo_begin: Vec<u16> = vec![0, 1, 3];
    let tap_set = TapSet {
        taps: Box::leak(taps.into_boxed_slice()),
        combo_taps: Box::leak(combo_taps.clone().into_boxed_slice()),
        combo_begin: Box::leak(combo_begin.clone().into_boxed_slice()),
        group_begin,
        combos_count: 2,
        reserved_register: 0,
        tot_combo_backs: 3,
    };
    let proof_system_info = *b"RISC0_STARK:v1__";
    let circuit_info = *b"RKVECTOR:v1_____";
    let globals = elems(n_globals, 41);
    let accum = elems(n * group_size[0], 42);
    let code = elems(n * group_size[1], 43);
    let data = elems(n * group_size[2], 44);
    let check = elems(4 * 4 * n, 45);

    let hashfn = &suite.hashfn;
    let mut prover = Prover::new(hal, Box::leak(Box::new(tap_set)));
    let enc = |s: &[u8; 16]| -> Vec<Elem> { s.iter().map(|&b| Elem::new(b as u32)).collect() }; // ProtocolInfo::encode
    prover.iop().commit(&hashfn.hash_elem_slice(&enc(&proof_system_info)));
    prover.iop().commit(&hashfn.hash_elem_slice(&enc(&circuit_info)));
    let mut io_po2 = globals.clone();
    io_po2.push(Elem::new(po2 as u32));
    prover.iop().commit(&hashfn.hash_elem_slice(&io_po2));
    prover.iop().write_field_elem_slice(&globals);
    prover.iop().write_u32_slice(&[po2 as u32]);
    prover.set_po2(po2);
    prover.commit_group(1, &hal.copy_from_elem("code", &code));
    prover.commit_group(2, &hal.copy_from_elem("data", &data));
    let mix: Vec<Elem> = (0..n_accum_mix).map(|_| prover.iop().random_elem()).collect();
    prover.commit_group(0, &hal.copy_from_elem("accum", &accum));
    let mix_buf = hal.copy_from_elem("mix", &mix);
    let io_buf = hal.copy_from_elem("io", &globals);
    let seal = prover.finalize(&[&mix_buf, &io_buf], &Given { check: check.clone() });

    let mut o = Out::new(4);
    for w in [po2, n_globals, n_accum_mix, group_size[0], group_size[1], group_size[2], reg_group.len(), 2, combo_taps.len()] {
        o.word(w as u32);
    }
    o.words(&reg_group);
    o.words(&reg_offset);
    o.words(&reg_combo);
    o.words(&combo_begin.iter().map(|&v| v as u32).collect::<Vec<_>>());
    o.words(&combo_taps.iter().map(|&v| v as u32).collect::<Vec<_>>());
    o.words(bytemuck::cast_slice::<u8, u32>(&proof_system_info));
    o.words(bytemuck::cast_slice::<u8, u32>(&circuit_info));
    o.elems(&globals);
    o.elems(&accum);
    o.elems(&code);
    o.elems(&data);
    o.elems(&check);
    o.word(seal.len() as u32);
    o.words(&seal);
    o.save(dir, "seal.bin")
}
