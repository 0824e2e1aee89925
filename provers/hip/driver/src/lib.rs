//! `HipProver`: raiko's `Prover` plugin backed by libraiko_hip.so (MI355X).
//!
//! Drop-in for `Risc0Prover` (provers/risc0/driver/src/lib.rs:51-122): same request object
//! (`config["risc0"]`, script/prove-block.sh:64-73), same `Proof` (hex of the journal), same cache
//! file (`bincode (String, Receipt)`, bonsai.rs:274-310).  What changes is the line
//! `session.prove()` (bonsai.rs:271): the executor and the rv32im circuit's witness / constraint
//! code stay risc0's, every prover stage behind `risc0_zkp::hal::Hal` runs in libraiko_hip.so
//! through the session entry point in its streaming form (`rk_stream_open / submit / wait / close`:
//! `rk_prove_session` for segments that arrive one by one), which keeps several segment proofs in
//! flight per GPU, stages the witness uploads ahead and verifies each seal, while `rk_stream_wait`
//! bounds how many witnesses exist at once; of the two circuit steps that need Fiat-Shamir
//! randomness `eval_check` runs on the GPU from the circuit's step list (`rk_program`) and
//! `accumulate` comes back as an `rk_circuit_hooks` callback.
//!
//! NOT COMPILED IN THE BUILD IMAGE (no Rust toolchain there).  Items of raiko itself are written
//! against the reference tree and cited; items *inside* risc0 crates (`Segment`, `CpuCircuitHal`,
//! the witness `Executor`/`Loader`, `SegmentReceipt` fields) are written from recollection of
//! risc0 1.0.1 and marked `RECALLED`: check them with `cargo check` first.
#![cfg(feature = "enable")]

pub mod ffi;
pub mod hal;

use std::{
    ffi::CStr,
    os::raw::{c_int, c_void},
    ptr,
};

use alloy_primitives::B256;
use hex::ToHex;
use raiko_lib::{
    input::{GuestInput, GuestOutput},
    prover::{IdStore, IdWrite, Proof, ProofKey, Prover, ProverConfig, ProverError, ProverResult},
};
use risc0_zkvm::{serde::to_vec, ExecutorEnv, ExecutorImpl, Receipt, Segment};
use serde::{Deserialize, Serialize};
use tracing::{error, info};

use crate::ffi::*;

/// The guest is raiko's risc0 guest: the ELF and image id come from the risc0 driver's generated
/// `methods` module (provers/risc0/driver/src/methods/risc0_guest.rs:1-5).
pub use risc0_driver_methods::{RISC0_GUEST_ELF, RISC0_GUEST_ID};
mod risc0_driver_methods {
    include!("../../../risc0/driver/src/methods/risc0_guest.rs");
}

/// Same option object as the risc0 driver (provers/risc0/driver/src/lib.rs:27-34).
#[derive(Clone, Debug, Serialize, Deserialize)]
pub struct Risc0Param {
    pub bonsai: bool,
    pub snark: bool,
    pub profile: bool,
    pub execution_po2: u32,
}

/// Optional `"hip": {...}` object of the request; every field has a default.
#[derive(Clone, Debug, Default, Deserialize)]
pub struct HipParam {
    /// GPUs of this node to use; empty = GPU 0.  All of them share one work queue of segments.
    #[serde(default)]
    pub devices: Vec<i32>,
    /// segment proofs in flight per GPU (1..=16; 3 saturates an MI355X at po2 = 20)
    #[serde(default = "default_inflight")]
    pub inflight: i32,
    /// staged witness uploads per GPU waiting for a prover (0..=16)
    #[serde(default = "default_upload_ahead")]
    pub upload_ahead: i32,
}
fn default_inflight() -> i32 {
    3
}
fn default_upload_ahead() -> i32 {
    2
}

pub struct HipProver;

/// Answers to `proof_type: "risc0"`, so it keeps risc0's prover code in proof keys
/// (provers/risc0/driver/src/lib.rs:53).
const HIP_PROVER_CODE: u8 = 3;

impl Prover for HipProver {
    async fn run(
        input: GuestInput,
        output: &GuestOutput,
        config: &ProverConfig,
        _store: Option<&mut dyn IdWrite>,
    ) -> ProverResult<Proof> {
        // lib.rs:63 unwraps; a missing or malformed option object is a Param error here
        let param: Risc0Param = serde_json::from_value(
            config.get("risc0").cloned().unwrap_or(serde_json::Value::Null),
        )?;
        let hip: HipParam = match config.get("hip") {
            Some(v) => serde_json::from_value(v.clone())?,
            None => HipParam { inflight: default_inflight(), upload_ahead: default_upload_ahead(), ..Default::default() },
        };
        if param.bonsai || param.snark {
            return Err(ProverError::GuestError(
                "the hip backend proves locally: bonsai / snark are not available".to_owned(),
            ));
        }
        let _proof_key: ProofKey = (input.chain_spec.chain_id, output.hash, HIP_PROVER_CODE);
        let encoded_input = to_vec(&input).map_err(|e| format!("Could not serialize proving input: {e}"))?; // lib.rs:71
        let expected = output.hash;

        // bonsai.rs:100-108: the cache label
        let encoded_output = to_vec(&expected).map_err(|e| e.to_string())?;
        let image_id = risc0_zkvm::compute_image_id(RISC0_GUEST_ELF).map_err(|e| e.to_string())?;
        let label = format!(
            "{}-{}",
            hex::encode(image_id),
            hex::encode(raiko_lib::primitives::keccak::keccak(bytemuck::cast_slice::<u32, u8>(&encoded_output)))
        );

        let receipt = match load_receipt(&label)? {
            Some((_uuid, receipt)) => receipt,
            None => {
                // the reference blocks the async worker for the whole proof (bonsai.rs:230 is a sync fn
                // called from async code); the SGX backend shows the remedy (provers/sgx/prover/src/lib.rs:276)
                let po2 = param.execution_po2;
                let profile = param.profile;
                #[cfg(feature = "pipelined")]
                let prove = prove_locally_pipelined;
                #[cfg(not(feature = "pipelined"))]
                let prove = prove_locally;
                let receipt = tokio::task::spawn_blocking(move || prove(po2, encoded_input, &hip, profile))
                    .await
                    .map_err(|e| ProverError::GuestError(e.to_string()))??;
                save_receipt(&label, &(String::new(), receipt.clone()))?;
                receipt
            }
        };
        // bonsai.rs:157-162: compared and logged, not fatal
        match receipt.journal.decode::<B256>() {
            Ok(got) if got == expected => info!("Prover succeeded"),
            other => error!("Output mismatch! Prover: {other:?}, expected: {expected:?}"),
        }
        Ok(Proof { proof: Some(receipt.journal.encode_hex()), quote: None, kzg_proof: None }) // lib.rs:84-111
    }

    /// A local proof cannot be interrupted and stores no remote id: a no-op like the SGX backend's
    /// (provers/sgx/prover/src/lib.rs:151-153).
    async fn cancel(_proof_key: ProofKey, _read: Box<&mut dyn IdStore>) -> ProverResult<()> {
        Ok(())
    }
}

// ------------------------------------------------------------------------------------------------
// prove_locally: bonsai.rs:230-272 with `session.prove()` replaced

/// One segment's witness as the circuit's CPU generator leaves it (column-major, Montgomery u32).
struct Witness {
    po2: u32,
    code: Vec<u32>,
    data: Vec<u32>,
    globals: Vec<u32>,
    /// the circuit hooks read `segment` again for the accum / check steps
    hook: Box<CircuitHook>,
}

/// Segments in flight between the witness generator and the GPU: an `rk_stream` with back-pressure.
///
/// The reference keeps a session's segments on disk (`segment_path`, bonsai.rs:261-266) so that a proof may be larger
/// than memory; a witness is ~1 GB at po2 = 20 and a block has hundreds to thousands of segments, so this side must
/// not hold more than a window of them either.  `submit` hands one witness to the stream and then blocks in
/// `rk_stream_wait` until at most `max_pending` = inflight + upload_ahead (per GPU) segments are unfinished; whatever
/// the finished prefix has released is dropped before the caller generates the next witness.  Seals (~0.3 MB each)
/// are the result and stay.  Dropping the window closes the stream on every path, also the early returns.
struct Window {
    stream: *mut rk_stream,
    device: c_int,
    max_pending: usize,
    taps: circuit::TapTables,
    // everything a submitted rk_segment points at lives until the stream reports it finished: boxed, never moved
    witnesses: std::collections::VecDeque<Option<Box<Witness>>>,
    first: usize, // submission index of witnesses[0]
    seals: Vec<Box<(Vec<u32>, usize)>>,
    // opts point into these: kept alive (and at a fixed address) for the life of the stream
    _devices: Vec<c_int>,
    _vopts: Box<rk_verify_opts>,
}

impl Window {
    fn open(hip: &HipParam) -> Result<Self, String> {
        let vopts = Box::new(rk_verify_opts {
            p2_rc_ext: ptr::null(),
            p2_rc_int: ptr::null(),
            p2_diag: ptr::null(),
            // the step list when the circuit crate exposes it (the identity is then checked by the library's own
            // evaluator), risc0's CircuitDef::poly_ext otherwise
            poly_ext: if circuit::program().is_null() { Some(circuit::poly_ext_trampoline) } else { None },
            user: ptr::null_mut(),
            program: circuit::program(),
            params: ptr::null(), // risc0's parameter set is the library default
        });
        let devices: Vec<c_int> = hip.devices.iter().map(|&d| d as c_int).collect();
        let opts = rk_session_opts {
            device: devices.first().copied().unwrap_or(0),
            inflight: hip.inflight,
            upload_ahead: hip.upload_ahead,
            verify: 1,
            devices: if devices.len() > 1 { devices.as_ptr() } else { ptr::null() },
            n_devices: if devices.len() > 1 { devices.len() as c_int } else { 0 },
            verify_opts: &*vopts,
            params: ptr::null(),
        };
        let mut stream: *mut rk_stream = ptr::null_mut();
        let st = unsafe { rk_stream_open(&opts, &mut stream) }; // devices / verify_opts are copied by the library
        if st != RK_OK {
            return Err(describe(st, opts.device, usize::MAX));
        }
        let gpus = devices.len().max(1);
        Ok(Window {
            stream,
            device: opts.device,
            max_pending: gpus * (hip.inflight.max(1) as usize + hip.upload_ahead.max(0) as usize),
            taps: circuit::tapset(),
            witnesses: Default::default(),
            first: 0,
            seals: Vec::new(),
            _devices: devices,
            _vopts: vopts,
        })
    }

    /// Hands one segment over; returns once the window has room for the next witness.
    fn submit(&mut self, w: Witness) -> Result<(), String> {
        let w = Box::new(w);
        let c_seg = w.as_rk_segment(&self.taps);
        let cap = unsafe { rk_seal_bound_words(&c_seg) };
        if cap == 0 {
            return Err(format!("segment {}: shape rejected by libraiko_hip", self.seals.len()));
        }
        let mut out = Box::new((vec![0u32; cap], 0usize));
        let st = unsafe { rk_stream_submit(self.stream, &c_seg, out.0.as_mut_ptr(), cap, &mut out.1) };
        // pushed before the status is looked at: the library may already hold pointers into both
        self.witnesses.push_back(Some(w));
        self.seals.push(out);
        if st != RK_OK {
            return Err(format!("rk_stream_submit: status {st}"));
        }
        let mut prefix = 0usize;
        let st = unsafe { rk_stream_wait(self.stream, self.max_pending, &mut prefix) };
        while self.first < prefix && !self.witnesses.is_empty() {
            self.witnesses.pop_front(); // ~1 GB each at po2 = 20: gone as soon as the seal has landed
            self.first += 1;
        }
        if st != RK_OK {
            return Err(describe(st, self.device, usize::MAX)); // finish() names the segment
        }
        Ok(())
    }

    /// Waits for everything submitted and returns the seals in submission order.
    fn finish(mut self) -> Result<Vec<Vec<u32>>, String> {
        let mut failed = usize::MAX;
        let st = unsafe { rk_stream_close(self.stream, &mut failed) };
        self.stream = ptr::null_mut();
        self.witnesses.clear();
        if st != RK_OK {
            return Err(describe(st, self.device, failed));
        }
        Ok(std::mem::take(&mut self.seals)
            .into_iter()
            .map(|b| {
                let (mut seal, w) = *b;
                seal.truncate(w);
                seal
            })
            .collect())
    }
}

impl Drop for Window {
    fn drop(&mut self) {
        if !self.stream.is_null() {
            // an early return (executor error, witness error): wait for what was submitted, then free the stream and
            // its worker thread; only after that may the witnesses and seal buffers go
            let mut failed = usize::MAX;
            unsafe { rk_stream_close(self.stream, &mut failed) };
            self.stream = ptr::null_mut();
        }
    }
}

fn executor_env<'a>(po2: u32, encoded_input: &'a [u32], dir: &std::path::Path, profile: bool) -> Result<ExecutorEnv<'a>, String> {
    // bonsai.rs:246-266 -- unchanged except for a private segment directory: the shared, wiped /tmp/risc0-cache of the
    // reference (bonsai.rs:261-265) races under concurrency_limit = 16
    let mut builder = ExecutorEnv::builder();
    builder.session_limit(None).segment_limit_po2(po2).write_slice(encoded_input);
    if profile {
        // bonsai.rs:252-255: the request's `profile: true` (script/prove-block.sh:64-73 always sends it)
        info!("Profiling enabled.");
        builder.enable_profiler("profile_r0_local.pb");
    }
    builder.segment_path(dir).build().map_err(|e| e.to_string())
}

fn prove_locally(po2: u32, encoded_input: Vec<u32>, hip: &HipParam, profile: bool) -> Result<Receipt, String> {
    let dir = tempfile::tempdir().map_err(|e| e.to_string())?;
    let env = executor_env(po2, &encoded_input, dir.path(), profile)?;
    let mut exec = ExecutorImpl::from_elf(env, RISC0_GUEST_ELF).map_err(|e| e.to_string())?;
    let session = exec.run().map_err(|e| e.to_string())?;
    // segments come back from disk one at a time; at most a window of witnesses exists at any moment
    let mut window = Window::open(hip)?; // after the executor: nothing to close if it fails
    for seg_ref in session.segments.iter() {
        let segment = seg_ref.resolve().map_err(|e| e.to_string())?;
        window.submit(circuit::witness(&segment)?)?;
    }
    let seals = window.finish()?;
    circuit::assemble_receipt(&session, seals)
}

/// The same with executor and prover overlapped (cargo feature `pipelined`): risc0's executor hands every
/// segment to a callback as it completes (RECALLED: `ExecutorImpl::run_with_callback`), the callback builds
/// the witness and submits it to the window, so segment k is proven while segment k + 1 executes and the
/// block's wall-clock is the executor's, not the sum.  The window's back-pressure bounds memory here too: a
/// callback that finds the window full blocks the executor until a proof has finished.
#[cfg(feature = "pipelined")]
fn prove_locally_pipelined(po2: u32, encoded_input: Vec<u32>, hip: &HipParam, profile: bool) -> Result<Receipt, String> {
    let dir = tempfile::tempdir().map_err(|e| e.to_string())?;
    let env = executor_env(po2, &encoded_input, dir.path(), profile)?;
    let mut exec = ExecutorImpl::from_elf(env, RISC0_GUEST_ELF).map_err(|e| e.to_string())?;
    let mut window = Window::open(hip)?; // after from_elf: an ELF that does not load leaves no stream behind
    let mut submit_err: Option<String> = None;
    let session = exec
        .run_with_callback(|segment| {
            if submit_err.is_none() {
                if let Err(e) = circuit::witness(&segment).and_then(|w| window.submit(w)) {
                    submit_err = Some(e);
                }
            }
            Ok(Box::new(risc0_zkvm::SimpleSegmentRef::new(segment)))
        })
        .map_err(|e| e.to_string());
    let seals = window.finish(); // waits for every submitted segment, also when the executor failed
    let session = session?;
    if let Some(e) = submit_err {
        return Err(e);
    }
    circuit::assemble_receipt(&session, seals?)
}

impl Witness {
    /// Pointers borrow `self`: the rk_segment must not outlive it.
    fn as_rk_segment(&self, taps: &circuit::TapTables) -> rk_segment {
        rk_segment {
            po2: self.po2,
            on_device: 0, // host arrays: rk_prove_session stages them onto the GPU that claims the segment
            taps: taps.as_rk_taps(),
            group: [ptr::null(), self.code.as_ptr(), self.data.as_ptr()], // accum comes from the hook
            check: ptr::null(),                                            // check comes from the hook
            globals: self.globals.as_ptr(),
            n_globals: self.globals.len() as u32,
            n_accum_mix: circuit::MIX_SIZE as u32,
            proof_system_info: *b"RISC0_STARK:v1__",
            circuit_info: *b"RV32IM:v1_______",
            hooks: &self.hook.hooks,
        }
    }
}

/// `rk_circuit_hooks` plus what its callbacks need; boxed so the address stays put.
pub struct CircuitHook {
    hooks: rk_circuit_hooks,
    steps: usize,
}

fn describe(st: c_int, device: c_int, failed: usize) -> String {
    let text = unsafe { CStr::from_ptr(rk_strerror(st)) }.to_string_lossy().into_owned();
    let detail = unsafe { CStr::from_ptr(rk_session_last_error(device)) }.to_string_lossy().into_owned();
    if failed == usize::MAX {
        format!("libraiko_hip: {text} ({detail})")
    } else {
        format!("libraiko_hip: {text} ({detail}), segment {failed}")
    }
}

// ------------------------------------------------------------------------------------------------
// receipt cache: bonsai.rs:274-310 with errors instead of expect()

fn zkp_cache_path(label: &str) -> std::path::PathBuf {
    std::env::temp_dir().join("raiko-hip-cache").join(format!("{label}.zkp"))
}

fn load_receipt(label: &str) -> ProverResult<Option<(String, Receipt)>> {
    if risc0_zkvm::is_dev_mode() {
        return Ok(None);
    }
    match std::fs::read(zkp_cache_path(label)) {
        Ok(raw) => bincode::deserialize(&raw).map(Some).map_err(|e| ProverError::GuestError(format!("cached receipt {label}: {e}"))),
        Err(_) => Ok(None),
    }
}

fn save_receipt(label: &str, data: &(String, Receipt)) -> ProverResult<()> {
    if risc0_zkvm::is_dev_mode() {
        return Ok(());
    }
    let path = zkp_cache_path(label);
    if let Some(dir) = path.parent() {
        std::fs::create_dir_all(dir)?;
    }
    let raw = bincode::serialize(data).map_err(|e| ProverError::GuestError(e.to_string()))?;
    std::fs::write(path, raw)?;
    Ok(())
}

// ------------------------------------------------------------------------------------------------
// The rv32im circuit's side: witness, accum, constraint polynomial -- risc0's own CPU code, called
// from the hooks on host copies of the device buffers.  A HIP `eval_check` generated from the
// circuit definition would replace the copies; until then this is the correct, slower form
// (about N * (Wc + Wd + Wa) * 4 * 5 bytes over PCIe per segment, hidden behind the other proofs in flight).
mod circuit {
    use super::*;
    // RECALLED (risc0-circuit-rv32im 1.0.1, risc0-zkp 1.0.1): module paths and signatures below
    use risc0_circuit_rv32im::{
        cpu::CpuCircuitHal,
        prove::{emu::preflight::PreflightTrace, witgen::WitnessGenerator},
        CircuitImpl, CIRCUIT, REGISTER_GROUP_ACCUM, REGISTER_GROUP_CODE, REGISTER_GROUP_DATA,
    };
    use risc0_zkp::{
        adapter::{CircuitInfo, PolyExt, TapsProvider},
        field::baby_bear::{BabyBearElem, BabyBearExtElem},
        hal::{cpu::CpuHal, CircuitHal, Hal},
        taps::TapSet,
    };

    pub const MIX_SIZE: usize = CircuitImpl::MIX_SIZE;

    /// The circuit's TapSet flattened into the arrays `rk_taps` points at.
    pub struct TapTables {
        group_size: [u32; 3],
        reg_group: Vec<u32>,
        reg_offset: Vec<u32>,
        reg_combo: Vec<u32>,
        combo_off: Vec<u32>,
        combo_backs: Vec<u32>,
    }
    impl TapTables {
        pub fn as_rk_taps(&self) -> rk_taps {
            rk_taps {
                group_size: self.group_size,
                n_regs: self.reg_group.len() as u32,
                reg_group: self.reg_group.as_ptr(),
                reg_offset: self.reg_offset.as_ptr(),
                reg_combo: self.reg_combo.as_ptr(),
                n_combos: (self.combo_off.len() - 1) as u32,
                combo_off: self.combo_off.as_ptr(),
                combo_backs: self.combo_backs.as_ptr(),
            }
        }
    }
    pub fn tapset() -> TapTables {
        let taps: &TapSet = CIRCUIT.get_taps();
        let mut t = TapTables {
            group_size: [
                taps.group_size(REGISTER_GROUP_ACCUM) as u32,
                taps.group_size(REGISTER_GROUP_CODE) as u32,
                taps.group_size(REGISTER_GROUP_DATA) as u32,
            ],
            reg_group: vec![],
            reg_offset: vec![],
            reg_combo: vec![],
            combo_off: vec![0],
            combo_backs: vec![],
        };
        for reg in taps.regs() {
            t.reg_group.push(reg.group() as u32);
            t.reg_offset.push(reg.offset() as u32);
            t.reg_combo.push(reg.combo_id() as u32);
        }
        for combo in taps.combos() {
            t.combo_backs.extend(combo.slice().iter().map(|&b| b as u32));
            t.combo_off.push(t.combo_backs.len() as u32);
        }
        t
    }

    /// The circuit's constraint polynomial as an `rk_program`, created once per process.
    /// RECALLED: risc0-circuit-rv32im 1.0.1 keeps the list as `poly_ext::DEF: PolyExtStepDef`
    /// (`block: &[PolyExtStep]`, `ret`), which `impl PolyExt for CircuitImpl` interprets; the module
    /// is private in the published crate, so this feature needs the one-line visibility patch
    /// (`pub mod poly_ext;`, provers/hip/patches/rv32im-poly-ext-pub.patch) on a vendored copy.  This is the default
    /// route: eval_check then runs on the GPU from the LDE the prover already holds and nothing crosses PCIe.  Null --
    /// and the CPU `eval_check` hook below takes over, downloading the 4.25 GB LDE per segment -- only when
    /// rk_program_create rejects the list or the crate is built with `--features cpu-eval-check` (an unpatched
    /// circuit crate).
    pub fn program() -> *const rk_program {
        #[cfg(not(feature = "cpu-eval-check"))]
        {
            use risc0_zkp::adapter::PolyExtStep as S;
            static PROGRAM: std::sync::OnceLock<usize> = std::sync::OnceLock::new();
            return *PROGRAM.get_or_init(|| {
                let def = &risc0_circuit_rv32im::poly_ext::DEF;
                let st = |op: rk_step_op, a: usize, b: usize, c: usize| rk_poly_step { op: op as u32, a: a as u32, b: b as u32, c: c as u32 };
                let steps: Vec<rk_poly_step> = def.block.iter().map(|s| match *s {
                    S::Const(v) => st(RK_STEP_CONST, v as usize, 0, 0),
                    S::Get(tap) => st(RK_STEP_GET, tap, 0, 0),
                    S::GetGlobal(base, off) => st(RK_STEP_GET_GLOBAL, base, off, 0),
                    S::Add(a, b) => st(RK_STEP_ADD, a, b, 0),
                    S::Sub(a, b) => st(RK_STEP_SUB, a, b, 0),
                    S::Mul(a, b) => st(RK_STEP_MUL, a, b, 0),
                    S::True => st(RK_STEP_TRUE, 0, 0, 0),
                    S::AndEqz(x, v) => st(RK_STEP_AND_EQZ, x, v, 0),
                    S::AndCond(x, cond, inner) => st(RK_STEP_AND_COND, x, cond, inner),
                }).collect();
                let taps = tapset();
                let mut prog: *mut rk_program = ptr::null_mut();
                let rc = unsafe { rk_program_create(steps.as_ptr(), steps.len(), def.ret as u32, &taps.as_rk_taps(), &mut prog) };
                // the generated kernel (hiprtc, once per process) instead of the interpreter, where it builds
                if rc == RK_OK {
                    let mut ctx: *mut rk_ctx = ptr::null_mut();
                    if unsafe { rk_ctx_create(0, ptr::null_mut(), &mut ctx) } == RK_OK {
                        let _ = unsafe { rk_program_compile(prog, ctx) };
                        unsafe { rk_ctx_destroy(ctx) };
                    }
                }
                if rc == RK_OK { prog as usize } else { 0 }
            }) as *const rk_program;
        }
        #[cfg(feature = "cpu-eval-check")]
        ptr::null()
    }

    /// Witness generation: what `SegmentProverImpl::prove_segment` does before its first commit.
    pub fn witness(segment: &Segment) -> Result<Witness, String> {
        let trace = PreflightTrace::new(segment).map_err(|e| e.to_string())?;
        let io = segment.prepare_globals();
        let witgen = WitnessGenerator::new(&CpuHal::new(risc0_zkp::core::hash::poseidon2::Poseidon2HashSuite::new_suite()),
                                           &CpuCircuitHal::new(), segment.po2, &io, trace);
        let steps = 1usize << segment.po2;
        let mut hook = Box::new(CircuitHook {
            // eval_check: with the step list (the default) the library evaluates the constraint polynomial on the
            // GPU from the LDE it already holds -- no download, no circuit-specific kernel; risc0's CPU evaluator
            // on host copies is the fallback when the list could not be created
            hooks: rk_circuit_hooks {
                user: ptr::null_mut(),
                accumulate: Some(accumulate),
                eval_check: if program().is_null() { Some(eval_check) } else { None },
                program: program(),
            },
            steps,
        });
        hook.hooks.user = &mut *hook as *mut CircuitHook as *mut c_void;
        Ok(Witness {
            po2: segment.po2 as u32,
            code: bytemuck::cast_slice(&witgen.code.to_vec()).to_vec(),
            data: bytemuck::cast_slice(&witgen.data.to_vec()).to_vec(),
            globals: bytemuck::cast_slice(&witgen.io.to_vec()).to_vec(),
            hook,
        })
    }

    fn download(view: &rk_circuit_view, d: *const u32, words: usize) -> Result<Vec<u32>, c_int> {
        let mut v = vec![0u32; words];
        let st = unsafe { rk_d2h(view.ctx, v.as_mut_ptr() as *mut c_void, d as *const c_void, words * 4) };
        if st == RK_OK { Ok(v) } else { Err(st) }
    }
    fn upload(view: &rk_circuit_view, d: *mut u32, v: &[u32]) -> c_int {
        unsafe { rk_h2d(view.ctx, d as *mut c_void, v.as_ptr() as *const c_void, v.len() * 4) }
    }

    /// `CircuitHal::accumulate(ctrl, io, data, mix, accum, steps)` on host copies.
    unsafe extern "C" fn accumulate(user: *mut c_void, view: *const rk_circuit_view, d_accum: *mut u32) -> c_int {
        let hook = &*(user as *const CircuitHook);
        let view = &*view;
        let run = || -> Result<(), c_int> {
            let hal = CpuHal::new(risc0_zkp::core::hash::poseidon2::Poseidon2HashSuite::new_suite());
            let n = hook.steps;
            let code = download(view, view.d_trace[1], n * view.group_size[1] as usize)?;
            let data = download(view, view.d_trace[2], n * view.group_size[2] as usize)?;
            let code = hal.copy_from_elem("code", bytemuck::cast_slice::<u32, BabyBearElem>(&code));
            let data = hal.copy_from_elem("data", bytemuck::cast_slice::<u32, BabyBearElem>(&data));
            let io = hal.copy_from_elem("io", bytemuck::cast_slice::<u32, BabyBearElem>(std::slice::from_raw_parts(view.globals, view.n_globals as usize)));
            let mix = hal.copy_from_elem("mix", bytemuck::cast_slice::<u32, BabyBearElem>(std::slice::from_raw_parts(view.mix, view.n_mix as usize)));
            let accum = hal.alloc_elem_init("accum", n * view.group_size[0] as usize, BabyBearElem::INVALID);
            CpuCircuitHal::new().accumulate(&code, &io, &data, &mix, &accum, n);
            hal.eltwise_zeroize_elem(&accum);
            let mut out = vec![0u32; accum.size()];
            accum.view(|s| out.copy_from_slice(bytemuck::cast_slice(s)));
            match upload(view, d_accum, &out) { RK_OK => Ok(()), st => Err(st) }
        };
        match std::panic::catch_unwind(std::panic::AssertUnwindSafe(run)) { Ok(Ok(())) => 0, Ok(Err(st)) => st, Err(_) => -1 }
    }

    /// `CircuitHal::eval_check(check, groups, globals, poly_mix, po2, steps)` on host copies.
    unsafe extern "C" fn eval_check(user: *mut c_void, view: *const rk_circuit_view, poly_mix: *const u32, d_check: *mut u32) -> c_int {
        let hook = &*(user as *const CircuitHook);
        let view = &*view;
        let run = || -> Result<(), c_int> {
            let hal = CpuHal::new(risc0_zkp::core::hash::poseidon2::Poseidon2HashSuite::new_suite());
            let domain = hook.steps * 4;
            let mut groups = Vec::with_capacity(3);
            for g in 0..3 {
                let host = download(view, view.d_lde[g], domain * view.group_size[g] as usize)?;
                groups.push(hal.copy_from_elem("lde", bytemuck::cast_slice::<u32, BabyBearElem>(&host)));
            }
            let io = hal.copy_from_elem("io", bytemuck::cast_slice::<u32, BabyBearElem>(std::slice::from_raw_parts(view.globals, view.n_globals as usize)));
            let mix = hal.copy_from_elem("mix", bytemuck::cast_slice::<u32, BabyBearElem>(std::slice::from_raw_parts(view.mix, view.n_mix as usize)));
            let check = hal.alloc_elem("check", 4 * domain);
            let pm = std::slice::from_raw_parts(poly_mix, 4);
            let poly_mix = BabyBearExtElem::from_subelems(bytemuck::cast_slice::<u32, BabyBearElem>(pm).iter().copied());
            let group_refs: Vec<&_> = groups.iter().collect();
            CpuCircuitHal::new().eval_check(&check, &group_refs, &[&mix, &io], poly_mix, view.po2 as usize, hook.steps);
            let mut out = vec![0u32; 4 * domain];
            check.view(|s| out.copy_from_slice(bytemuck::cast_slice(s)));
            match upload(view, d_check, &out) { RK_OK => Ok(()), st => Err(st) }
        };
        match std::panic::catch_unwind(std::panic::AssertUnwindSafe(run)) { Ok(Ok(())) => 0, Ok(Err(st)) => st, Err(_) => -1 }
    }

    /// `CircuitDef::poly_ext` for rk_verify_segment_ex: the verifier's constraint identity.
    pub unsafe extern "C" fn poly_ext_trampoline(
        _user: *mut c_void, pub_: *const rk_segment, poly_mix: *const u32, eval_u_ext: *const u32, n_taps: usize,
        mix: *const u32, n_mix: u32, out_ext: *mut u32,
    ) -> c_int {
        let seg = &*pub_;
        let pm = BabyBearExtElem::from_subelems(bytemuck::cast_slice::<u32, BabyBearElem>(std::slice::from_raw_parts(poly_mix, 4)).iter().copied());
        let eval_u: &[BabyBearExtElem] = bytemuck::cast_slice(std::slice::from_raw_parts(eval_u_ext, 4 * n_taps));
        let out: &[BabyBearElem] = bytemuck::cast_slice(std::slice::from_raw_parts(seg.globals, seg.n_globals as usize));
        let mix: &[BabyBearElem] = bytemuck::cast_slice(std::slice::from_raw_parts(mix, n_mix as usize));
        let tot = CIRCUIT.poly_ext(&pm, eval_u, &[out, mix]).tot;
        let words: &[u32] = bytemuck::cast_slice(tot.subelems());
        std::ptr::copy_nonoverlapping(words.as_ptr(), out_ext, 4);
        0
    }

    /// `CompositeReceipt` from the seals: what `session.prove()` returns besides proving.
    pub fn assemble_receipt(session: &risc0_zkvm::Session, seals: Vec<Vec<u32>>) -> Result<Receipt, String> {
        use risc0_zkvm::{CompositeReceipt, InnerReceipt, SegmentReceipt};
        let mut segments = Vec::with_capacity(seals.len());
        for (index, (seal, seg_ref)) in seals.into_iter().zip(session.segments.iter()).enumerate() {
            let segment = seg_ref.resolve().map_err(|e| e.to_string())?;
            let claim = risc0_zkvm::ReceiptClaim::decode(&seal).map_err(|e| e.to_string())?; // the claim is read back from the seal's globals
            segments.push(SegmentReceipt {
                seal,
                index: index as u32,
                hashfn: "poseidon2".to_owned(),
                verifier_parameters: risc0_zkvm::SegmentReceiptVerifierParameters::default().digest(),
                claim,
            });
            drop(segment);
        }
        let composite = CompositeReceipt {
            segments,
            assumption_receipts: vec![],
            verifier_parameters: risc0_zkvm::CompositeReceiptVerifierParameters::default().digest(),
        };
        Ok(Receipt::new(InnerReceipt::Composite(composite), session.journal.clone().map(|j| j.bytes).unwrap_or_default()))
    }
}

#[cfg(test)]
mod test {
    use super::*;

    /// The reference's own smoke test shape (provers/risc0/driver/src/lib.rs:131-137): prove, then verify.
    #[test]
    fn abi_matches_and_a_gpu_is_visible() {
        assert_eq!(unsafe { rk_abi_version() }, 4);
        let mut n = 0;
        assert_eq!(unsafe { rk_device_count(&mut n) }, RK_OK);
        assert!(n > 0, "no MI355X visible: the hip backend has no CPU fallback");
    }
}
