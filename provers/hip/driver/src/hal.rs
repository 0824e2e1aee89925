//! `impl risc0_zkp::hal::Hal for HipHal`: the operator-level route.  risc0's own generic prover
//! (`risc0_zkp::prove::Prover<H>` driven by `risc0_circuit_rv32im::prove::SegmentProverImpl<H, C>`)
//! then runs unchanged and every `Hal` call lands on one `rk_*` entry point -- the same split
//! risc0's CUDA and Metal backends use.  `HipProver` (lib.rs) takes the whole-segment route
//! (`rk_prove_session` + circuit hooks) instead, which keeps several proofs in flight; this module
//! is for callers that want risc0's prover logic and only the kernels from libraiko_hip.so.
//!
//! RECALLED: the trait below is written from recollection of risc0-zkp 1.0.1 `hal/mod.rs`
//! (method names, argument order, the `Buffer` trait).  `cargo check` against the real crate is
//! the first step after vendoring this file.
use std::{cell::RefCell, fmt::Debug, marker::PhantomData, os::raw::c_void, ptr, rc::Rc};

use risc0_core::field::baby_bear::{BabyBear, BabyBearElem, BabyBearExtElem};
use risc0_zkp::{
    core::{digest::Digest, hash::HashSuite},
    hal::{Buffer, Hal},
};

use crate::ffi::*;

fn ck(ctx: *mut rk_ctx, st: i32, what: &str) {
    if st != RK_OK {
        let detail = unsafe { std::ffi::CStr::from_ptr(rk_last_error(ctx)) }.to_string_lossy().into_owned();
        // the Hal trait has no error channel (risc0's own backends panic on a failed launch too)
        panic!("libraiko_hip: {what} failed with status {st} ({detail})");
    }
}

struct Ctx(*mut rk_ctx);
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { rk_ctx_destroy(self.0) };
    }
}

/// Device allocation + element range; clones share the allocation like risc0's own buffers.
struct Alloc {
    ctx: Rc<Ctx>,
    ptr: *mut c_void,
}
impl Drop for Alloc {
    fn drop(&mut self) {
        unsafe { rk_free(self.ctx.0, self.ptr) };
    }
}

#[derive(Clone)]
pub struct HipBuffer<T> {
    name: &'static str,
    alloc: Rc<Alloc>,
    offset: usize, // in elements of T
    size: usize,
    marker: PhantomData<T>,
}

impl<T> HipBuffer<T> {
    fn words_per_elem() -> usize {
        std::mem::size_of::<T>() / 4
    }
    fn new(ctx: &Rc<Ctx>, name: &'static str, size: usize) -> Self {
        let mut p = ptr::null_mut();
        ck(ctx.0, unsafe { rk_alloc(ctx.0, size * std::mem::size_of::<T>(), &mut p) }, "rk_alloc");
        Self { name, alloc: Rc::new(Alloc { ctx: ctx.clone(), ptr: p }), offset: 0, size, marker: PhantomData }
    }
    /// device address of element 0 of this (sub)buffer as u32 words
    pub fn as_ptr(&self) -> *mut u32 {
        unsafe { (self.alloc.ptr as *mut u32).add(self.offset * Self::words_per_elem()) }
    }
    fn ctx(&self) -> *mut rk_ctx {
        self.alloc.ctx.0
    }
}

impl<T: Clone + bytemuck::Pod> Buffer<T> for HipBuffer<T> {
    fn name(&self) -> &'static str {
        self.name
    }
    fn size(&self) -> usize {
        self.size
    }
    fn slice(&self, offset: usize, size: usize) -> Self {
        assert!(offset + size <= self.size);
        Self { name: self.name, alloc: self.alloc.clone(), offset: self.offset + offset, size, marker: PhantomData }
    }
    fn view<F: FnOnce(&[T])>(&self, f: F) {
        let mut host = vec![T::zeroed(); self.size];
        ck(self.ctx(), unsafe { rk_d2h(self.ctx(), host.as_mut_ptr() as *mut c_void, self.as_ptr() as *const c_void, self.size * std::mem::size_of::<T>()) }, "rk_d2h");
        f(&host)
    }
    fn view_mut<F: FnOnce(&mut [T])>(&self, f: F) {
        let bytes = self.size * std::mem::size_of::<T>();
        let mut host = vec![T::zeroed(); self.size];
        ck(self.ctx(), unsafe { rk_d2h(self.ctx(), host.as_mut_ptr() as *mut c_void, self.as_ptr() as *const c_void, bytes) }, "rk_d2h");
        f(&mut host);
        ck(self.ctx(), unsafe { rk_h2d(self.ctx(), self.as_ptr() as *mut c_void, host.as_ptr() as *const c_void, bytes) }, "rk_h2d");
    }
}

pub struct HipHal {
    ctx: Rc<Ctx>,
    suite: HashSuite<BabyBear>,
    /// kept so `rk_mix_poly_coeffs` / `rk_batch_evaluate_any` can take their small index arrays from the host
    scratch: RefCell<Vec<u32>>,
}

impl HipHal {
    /// One context (HIP stream + scratch pool) on `device`; the Poseidon2 suite is the default one.
    pub fn new(device: i32) -> Self {
        let mut ctx = ptr::null_mut();
        let st = unsafe { rk_ctx_create(device, ptr::null_mut(), &mut ctx) };
        assert_eq!(st, RK_OK, "rk_ctx_create({device}) = {st}: the hip backend has no CPU fallback");
        Self {
            ctx: Rc::new(Ctx(ctx)),
            suite: risc0_zkp::core::hash::poseidon2::Poseidon2HashSuite::new_suite(),
            scratch: RefCell::new(vec![]),
        }
    }
    fn raw(&self) -> *mut rk_ctx {
        self.ctx.0
    }
    fn upload<T: bytemuck::Pod>(&self, name: &'static str, slice: &[T]) -> HipBuffer<T> {
        let buf = HipBuffer::<T>::new(&self.ctx, name, slice.len());
        ck(self.raw(), unsafe { rk_h2d(self.raw(), buf.as_ptr() as *mut c_void, slice.as_ptr() as *const c_void, std::mem::size_of_val(slice)) }, "rk_h2d");
        buf
    }
}

impl Hal for HipHal {
    type Field = BabyBear;
    type Elem = BabyBearElem;
    type ExtElem = BabyBearExtElem;
    type Buffer<T: Clone + Debug + PartialEq> = HipBuffer<T>;

    fn has_unified_memory(&self) -> bool {
        false
    }
    fn get_hash_suite(&self) -> &HashSuite<Self::Field> {
        &self.suite
    }

    fn alloc_digest(&self, name: &'static str, size: usize) -> Self::Buffer<Digest> {
        HipBuffer::new(&self.ctx, name, size)
    }
    fn alloc_elem(&self, name: &'static str, size: usize) -> Self::Buffer<Self::Elem> {
        HipBuffer::new(&self.ctx, name, size)
    }
    fn alloc_elem_init(&self, name: &'static str, size: usize, value: Self::Elem) -> Self::Buffer<Self::Elem> {
        self.upload(name, &vec![value; size])
    }
    fn alloc_extelem(&self, name: &'static str, size: usize) -> Self::Buffer<Self::ExtElem> {
        HipBuffer::new(&self.ctx, name, size)
    }
    fn alloc_u32(&self, name: &'static str, size: usize) -> Self::Buffer<u32> {
        HipBuffer::new(&self.ctx, name, size)
    }
    fn copy_from_digest(&self, name: &'static str, slice: &[Digest]) -> Self::Buffer<Digest> {
        self.upload(name, slice)
    }
    fn copy_from_elem(&self, name: &'static str, slice: &[Self::Elem]) -> Self::Buffer<Self::Elem> {
        self.upload(name, slice)
    }
    fn copy_from_extelem(&self, name: &'static str, slice: &[Self::ExtElem]) -> Self::Buffer<Self::ExtElem> {
        self.upload(name, slice)
    }
    fn copy_from_u32(&self, name: &'static str, slice: &[u32]) -> Self::Buffer<u32> {
        self.upload(name, slice)
    }

    fn batch_expand_into_evaluate_ntt(&self, output: &Self::Buffer<Self::Elem>, input: &Self::Buffer<Self::Elem>, count: usize, expand_bits: usize) {
        let in_size = input.size() / count;
        assert_eq!(output.size(), input.size() << expand_bits);
        ck(self.raw(), unsafe { rk_batch_expand_into_evaluate_ntt(self.raw(), output.as_ptr(), input.as_ptr(), in_size, count, expand_bits as u32) }, "rk_batch_expand_into_evaluate_ntt");
    }
    fn batch_interpolate_ntt(&self, io: &Self::Buffer<Self::Elem>, count: usize) {
        ck(self.raw(), unsafe { rk_batch_interpolate_ntt(self.raw(), io.as_ptr(), io.size() / count, count) }, "rk_batch_interpolate_ntt");
    }
    fn batch_bit_reverse(&self, io: &Self::Buffer<Self::Elem>, count: usize) {
        ck(self.raw(), unsafe { rk_batch_bit_reverse(self.raw(), io.as_ptr(), io.size() / count, count) }, "rk_batch_bit_reverse");
    }
    fn batch_evaluate_any(&self, coeffs: &Self::Buffer<Self::Elem>, poly_count: usize, which: &Self::Buffer<u32>, xs: &Self::Buffer<Self::ExtElem>, out: &Self::Buffer<Self::ExtElem>) {
        // which / xs / out are transcript-sized: the C entry point takes them from the host
        let size = coeffs.size() / poly_count;
        let mut h_which = vec![];
        which.view(|w| h_which.extend_from_slice(w));
        let mut h_xs: Vec<u32> = vec![];
        xs.view(|x| h_xs.extend_from_slice(bytemuck::cast_slice(x)));
        let mut h_out = vec![0u32; 4 * h_which.len()];
        ck(self.raw(), unsafe { rk_batch_evaluate_any(self.raw(), coeffs.as_ptr(), poly_count, size, h_which.as_ptr(), h_xs.as_ptr(), h_which.len(), h_out.as_mut_ptr()) }, "rk_batch_evaluate_any");
        ck(self.raw(), unsafe { rk_h2d(self.raw(), out.as_ptr() as *mut c_void, h_out.as_ptr() as *const c_void, h_out.len() * 4) }, "rk_h2d");
    }
    fn zk_shift(&self, io: &Self::Buffer<Self::Elem>, count: usize) {
        ck(self.raw(), unsafe { rk_zk_shift(self.raw(), io.as_ptr(), io.size() / count, count) }, "rk_zk_shift");
    }
    fn mix_poly_coeffs(&self, output: &Self::Buffer<Self::ExtElem>, mix_start: &Self::ExtElem, mix: &Self::ExtElem, input: &Self::Buffer<Self::Elem>, combos: &Self::Buffer<u32>, input_size: usize, count: usize) {
        let mut scratch = self.scratch.borrow_mut();
        scratch.clear();
        combos.view(|c| scratch.extend_from_slice(c));
        let ms: &[u32] = bytemuck::cast_slice(std::slice::from_ref(mix_start));
        let mx: &[u32] = bytemuck::cast_slice(std::slice::from_ref(mix));
        ck(self.raw(), unsafe { rk_mix_poly_coeffs(self.raw(), output.as_ptr(), ms.as_ptr(), mx.as_ptr(), input.as_ptr(), scratch.as_ptr(), input_size, count) }, "rk_mix_poly_coeffs");
    }
    fn eltwise_add_elem(&self, output: &Self::Buffer<Self::Elem>, input1: &Self::Buffer<Self::Elem>, input2: &Self::Buffer<Self::Elem>) {
        ck(self.raw(), unsafe { rk_eltwise_add_elem(self.raw(), output.as_ptr(), input1.as_ptr(), input2.as_ptr(), output.size()) }, "rk_eltwise_add_elem");
    }
    fn eltwise_sum_extelem(&self, output: &Self::Buffer<Self::Elem>, input: &Self::Buffer<Self::ExtElem>) {
        let count = output.size() / 4;
        ck(self.raw(), unsafe { rk_eltwise_sum_extelem(self.raw(), output.as_ptr(), input.as_ptr(), count, input.size() / count) }, "rk_eltwise_sum_extelem");
    }
    fn eltwise_copy_elem(&self, output: &Self::Buffer<Self::Elem>, input: &Self::Buffer<Self::Elem>) {
        ck(self.raw(), unsafe { rk_eltwise_copy_elem(self.raw(), output.as_ptr(), input.as_ptr(), output.size()) }, "rk_eltwise_copy_elem");
    }
    fn eltwise_zeroize_elem(&self, elems: &Self::Buffer<Self::Elem>) {
        ck(self.raw(), unsafe { rk_eltwise_zeroize_elem(self.raw(), elems.as_ptr(), elems.size()) }, "rk_eltwise_zeroize_elem");
    }
    fn fri_fold(&self, output: &Self::Buffer<Self::Elem>, input: &Self::Buffer<Self::Elem>, mix: &Self::ExtElem) {
        let mx: &[u32] = bytemuck::cast_slice(std::slice::from_ref(mix));
        ck(self.raw(), unsafe { rk_fri_fold(self.raw(), output.as_ptr(), input.as_ptr(), output.size() / 4, mx.as_ptr()) }, "rk_fri_fold");
    }
    fn hash_rows(&self, output: &Self::Buffer<Digest>, matrix: &Self::Buffer<Self::Elem>) {
        let rows = output.size();
        ck(self.raw(), unsafe { rk_hash_rows(self.raw(), output.as_ptr(), matrix.as_ptr(), rows, matrix.size() / rows) }, "rk_hash_rows");
    }
    fn hash_fold(&self, io: &Self::Buffer<Digest>, input_size: usize, output_size: usize) {
        ck(self.raw(), unsafe { rk_hash_fold(self.raw(), io.as_ptr(), input_size, output_size) }, "rk_hash_fold");
    }
    fn gather_sample(&self, dst: &Self::Buffer<Self::Elem>, src: &Self::Buffer<Self::Elem>, idx: usize, size: usize, stride: usize) {
        ck(self.raw(), unsafe { rk_gather_sample(self.raw(), dst.as_ptr(), src.as_ptr(), idx, size, stride) }, "rk_gather_sample");
    }
    fn prefix_products(&self, io: &Self::Buffer<Self::ExtElem>) {
        ck(self.raw(), unsafe { rk_prefix_products(self.raw(), io.as_ptr(), io.size()) }, "rk_prefix_products");
    }
    fn scatter(&self, into: &Self::Buffer<Self::Elem>, index: &[u32], offsets: &[u32], values: &[Self::Elem]) {
        if index.is_empty() {
            return;
        }
        let vals: &[u32] = bytemuck::cast_slice(values);
        ck(self.raw(), unsafe { rk_scatter(self.raw(), into.as_ptr(), into.size(), index.as_ptr(), index.len() - 1, offsets.as_ptr(), vals.as_ptr()) }, "rk_scatter");
    }
}

// ------------------------------------------------------------------------------------------------
// `impl CircuitHal<HipHal>`: the circuit's half of the operator route.  risc0's CUDA / Metal back ends
// ship a generated `eval_check` kernel per circuit; here the circuit's constraint list
// (`PolyExtStepDef`) is handed to the library once (`rk_program_create`) and `eval_check` runs it on
// the buffers the prover already holds on the GPU (`rk_program_eval_check`).  `accumulate` stays
// risc0's CPU code on host views of the device buffers.
//
// RECALLED: `risc0_zkp::hal::CircuitHal` of 1.0.1 -- `eval_check(check, groups, globals, poly_mix, po2,
// steps)` with groups in REGISTER_GROUP order (accum, code, data) and globals = [mix, out];
// `accumulate(ctrl, io, data, mix, accum, steps)`.
pub struct HipCircuitHal {
    program: *mut rk_program,
    cpu: risc0_circuit_rv32im::cpu::CpuCircuitHal,
}

impl HipCircuitHal {
    /// `steps` / `ret`: the circuit's `PolyExtStepDef` flattened to `rk_poly_step`s (see
    /// `crate::circuit::program` in lib.rs for the conversion); `taps`: its TapSet as `rk_taps`.
    pub fn new(steps: &[rk_poly_step], ret: u32, taps: &rk_taps) -> Result<Self, i32> {
        let mut program = ptr::null_mut();
        let st = unsafe { rk_program_create(steps.as_ptr(), steps.len(), ret, taps, &mut program) };
        if st != RK_OK {
            return Err(st);
        }
        Ok(Self { program, cpu: risc0_circuit_rv32im::cpu::CpuCircuitHal::new() })
    }
}

impl Drop for HipCircuitHal {
    fn drop(&mut self) {
        unsafe { rk_program_destroy(self.program) };
    }
}

impl risc0_zkp::hal::CircuitHal<HipHal> for HipCircuitHal {
    fn eval_check(
        &self,
        check: &HipBuffer<BabyBearElem>,
        groups: &[&HipBuffer<BabyBearElem>],
        globals: &[&HipBuffer<BabyBearElem>],
        poly_mix: BabyBearExtElem,
        po2: usize,
        steps: usize,
    ) {
        let ctx = check.ctx();
        let domain = steps * 4;
        // globals = [mix, out]: small, read back once per proof
        let mut mix = vec![0u32; globals[0].size()];
        let mut out = vec![0u32; globals[1].size()];
        globals[0].view(|s| mix.copy_from_slice(bytemuck::cast_slice(s)));
        globals[1].view(|s| out.copy_from_slice(bytemuck::cast_slice(s)));
        let view = rk_circuit_view {
            ctx,
            stream: ptr::null_mut(), // the library uses the context's own stream
            po2: po2 as u32,
            group_size: [
                (groups[0].size() / domain) as u32,
                (groups[1].size() / domain) as u32,
                (groups[2].size() / domain) as u32,
            ],
            d_trace: [ptr::null(); 3],
            d_lde: [groups[0].as_ptr() as *const u32, groups[1].as_ptr() as *const u32, groups[2].as_ptr() as *const u32],
            globals: out.as_ptr(),
            n_globals: out.len() as u32,
            mix: mix.as_ptr(),
            n_mix: mix.len() as u32,
        };
        let pm: [u32; 4] = bytemuck::cast(poly_mix);
        ck(ctx, unsafe { rk_program_eval_check(self.program, &view, pm.as_ptr(), check.as_ptr()) }, "rk_program_eval_check");
    }

    fn accumulate(
        &self,
        ctrl: &HipBuffer<BabyBearElem>,
        io: &HipBuffer<BabyBearElem>,
        data: &HipBuffer<BabyBearElem>,
        mix: &HipBuffer<BabyBearElem>,
        accum: &HipBuffer<BabyBearElem>,
        steps: usize,
    ) {
        // risc0's CPU accumulate on host copies (the buffers' view / view_mut round-trip through rk_d2h / rk_h2d)
        let cpu_hal = risc0_zkp::hal::cpu::CpuHal::new(risc0_zkp::core::hash::poseidon2::Poseidon2HashSuite::new_suite());
        let host = |b: &HipBuffer<BabyBearElem>, name: &'static str| {
            let mut v = vec![BabyBearElem::default(); b.size()];
            b.view(|s| v.copy_from_slice(s));
            cpu_hal.copy_from_elem(name, &v)
        };
        let (c, i, d, m, a) = (host(ctrl, "ctrl"), host(io, "io"), host(data, "data"), host(mix, "mix"), host(accum, "accum"));
        risc0_zkp::hal::CircuitHal::accumulate(&self.cpu, &c, &i, &d, &m, &a, steps);
        a.view(|s| accum.view_mut(|dst| dst.copy_from_slice(s)));
    }
}
