//! `extern "C"` declarations of libraiko_hip.so -- the mechanical translation of
//! `include/raiko_hip.h` (the C header is the source of truth; `tools/check_ffi.py` regenerates
//! this body with `--emit` and the CPU test suite fails when the two drift apart).
//!
//! Reference interfaces these entry points stand behind: `trait Prover` (lib/src/prover.rs:52-62),
//! `prove_locally` -> `session.prove()` (provers/risc0/driver/src/bonsai.rs:230-272) and, one level
//! down, `risc0_zkp::hal::{Hal, CircuitHal}` of risc0-zkp 1.0.1 (Cargo.lock:7243).
#![allow(non_camel_case_types)]
#![allow(clippy::missing_safety_doc)]
use std::os::raw::{c_char, c_int, c_void};
pub const RK_ECALL_HALT: c_int = 0;
pub const RK_ECALL_READ: c_int = 1;
pub const RK_ECALL_COMMIT: c_int = 2;

pub const RK_EXIT_HALTED: c_int = 0;
pub const RK_EXIT_SYSTEM_SPLIT: c_int = 2;

pub type rk_status = c_int;
pub const RK_OK: rk_status = 0;
pub const RK_ERR_INVALID: rk_status = -1;
pub const RK_ERR_HIP: rk_status = -2;
pub const RK_ERR_NOMEM: rk_status = -3;
pub const RK_ERR_NODEVICE: rk_status = -4;
pub const RK_ERR_CAPACITY: rk_status = -5;
pub const RK_ERR_INTERNAL: rk_status = -6;
pub const RK_ERR_VERIFY: rk_status = -7;
pub const RK_ERR_CALLBACK: rk_status = -8;

pub type rk_preset = c_int;
pub const RK_PRESET_RISC0: rk_preset = 0;
pub const RK_PRESET_SP1: rk_preset = 1;

pub type rk_step_op = c_int;
pub const RK_STEP_CONST: rk_step_op = 0;
pub const RK_STEP_GET: rk_step_op = 1;
pub const RK_STEP_GET_GLOBAL: rk_step_op = 2;
pub const RK_STEP_ADD: rk_step_op = 3;
pub const RK_STEP_SUB: rk_step_op = 4;
pub const RK_STEP_MUL: rk_step_op = 5;
pub const RK_STEP_TRUE: rk_step_op = 6;
pub const RK_STEP_AND_EQZ: rk_step_op = 7;
pub const RK_STEP_AND_COND: rk_step_op = 8;

pub type rk_air_op = c_int;
pub const RK_AIR_CONST: rk_air_op = 0;
pub const RK_AIR_LOCAL: rk_air_op = 1;
pub const RK_AIR_NEXT: rk_air_op = 2;
pub const RK_AIR_PUBLIC: rk_air_op = 3;
pub const RK_AIR_IS_FIRST_ROW: rk_air_op = 4;
pub const RK_AIR_IS_LAST_ROW: rk_air_op = 5;
pub const RK_AIR_IS_TRANSITION: rk_air_op = 6;
pub const RK_AIR_ADD: rk_air_op = 7;
pub const RK_AIR_SUB: rk_air_op = 8;
pub const RK_AIR_MUL: rk_air_op = 9;
pub const RK_AIR_NEG: rk_air_op = 10;
pub const RK_AIR_ASSERT_ZERO: rk_air_op = 11;
pub const RK_AIR_PERM_LOCAL: rk_air_op = 12;
pub const RK_AIR_PERM_NEXT: rk_air_op = 13;
pub const RK_AIR_CHALLENGE: rk_air_op = 14;
pub const RK_AIR_CUMSUM: rk_air_op = 15;

pub type rk_kclass = c_int;
pub const RK_KCLASS_HASH_ROWS: rk_kclass = 0;
pub const RK_KCLASS_HASH_FOLD: rk_kclass = 1;
pub const RK_KCLASS_NTT_PASS: rk_kclass = 2;
pub const RK_KCLASS_BIT_REVERSE: rk_kclass = 3;
pub const RK_KCLASS_POLY: rk_kclass = 4;
pub const RK_KCLASS_COUNT: rk_kclass = 5;

pub const RK_MAX_QUERIES: u32 = 256;
pub const RK_COMM_ID_BYTES: u32 = 128;
pub const RK_TRACE_CODE_COLS: u32 = 2;
pub const RK_TRACE_DATA_COLS: u32 = 16;

#[repr(C)]
pub struct rk_air {
    _private: [u8; 0],
}

#[repr(C)]
pub struct rk_comm {
    _private: [u8; 0],
}

#[repr(C)]
pub struct rk_ctx {
    _private: [u8; 0],
}

#[repr(C)]
pub struct rk_exec {
    _private: [u8; 0],
}

#[repr(C)]
pub struct rk_program {
    _private: [u8; 0],
}

#[repr(C)]
pub struct rk_stream {
    _private: [u8; 0],
}

pub type rk_poly_ext_fn = unsafe extern "C" fn(user: *mut c_void, pub_: *const rk_segment, poly_mix: *const u32, eval_u_ext: *const u32, n_taps: usize, mix: *const u32, n_mix: u32, out_ext: *mut u32) -> c_int;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_params {
    pub struct_size: u32,
    pub ext_w: u32,
    pub root_2_27: u32,
    pub coset_shift: u32,
    pub p2_width: u32,
    pub p2_m4: u32,
    pub p2_pad_free: u32,
    pub p2_rc_ext: *const u32,
    pub p2_rc_int: *const u32,
    pub p2_diag: *const u32,
    pub queries: u32,
    pub blowup_log2: u32,
    pub fri_fold_log2: u32,
    pub fri_min_degree: u32,
    pub pow_bits: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_matrix {
    pub d_values: *const u32,
    pub height: u32,
    pub width: u32,
    pub row_major: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_taps {
    pub group_size: [u32; 3],
    pub n_regs: u32,
    pub reg_group: *const u32,
    pub reg_offset: *const u32,
    pub reg_combo: *const u32,
    pub n_combos: u32,
    pub combo_off: *const u32,
    pub combo_backs: *const u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_circuit_view {
    pub ctx: *mut rk_ctx,
    pub stream: *mut c_void,
    pub po2: u32,
    pub group_size: [u32; 3],
    pub d_trace: [*const u32; 3],
    pub d_lde: [*const u32; 3],
    pub globals: *const u32,
    pub n_globals: u32,
    pub mix: *const u32,
    pub n_mix: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_circuit_hooks {
    pub user: *mut c_void,
    pub accumulate: Option<unsafe extern "C" fn(user: *mut c_void, view: *const rk_circuit_view, d_accum: *mut u32) -> c_int>,
    pub eval_check: Option<unsafe extern "C" fn(user: *mut c_void, view: *const rk_circuit_view, poly_mix: *const u32, d_check: *mut u32) -> c_int>,
    pub program: *const rk_program,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_poly_step {
    pub op: u32,
    pub a: u32,
    pub b: u32,
    pub c: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_program_info {
    pub n_steps: u64,
    pub n_ops: u64,
    pub n_fp_slots: u32,
    pub n_mix_slots: u32,
    pub n_consts: u32,
    pub n_mix_powers: u32,
    pub max_power: u32,
    pub n_taps: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_segment {
    pub po2: u32,
    pub on_device: u32,
    pub taps: rk_taps,
    pub group: [*const u32; 3],
    pub check: *const u32,
    pub globals: *const u32,
    pub n_globals: u32,
    pub n_accum_mix: u32,
    pub proof_system_info: [u8; 16],
    pub circuit_info: [u8; 16],
    pub hooks: *const rk_circuit_hooks,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_verify_opts {
    pub p2_rc_ext: *const u32,
    pub p2_rc_int: *const u32,
    pub p2_diag: *const u32,
    pub poly_ext: Option<rk_poly_ext_fn>,
    pub user: *mut c_void,
    pub program: *const rk_program,
    pub params: *const rk_params,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_session_opts {
    pub device: c_int,
    pub inflight: c_int,
    pub upload_ahead: c_int,
    pub verify: c_int,
    pub devices: *const c_int,
    pub n_devices: c_int,
    pub verify_opts: *const rk_verify_opts,
    pub params: *const rk_params,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_exec_opts {
    pub struct_size: u32,
    pub segment_limit_po2: u32,
    pub session_limit: u64,
    pub input_words: *const u32,
    pub n_input_words: usize,
    pub record_trace: u32,
    pub profile: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_exec_summary {
    pub total_cycles: u64,
    pub n_segments: u32,
    pub exit_code: u32,
    pub journal_bytes: usize,
    pub input_words_read: usize,
    pub status: c_int,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_exec_segment {
    pub index: u32,
    pub po2: u32,
    pub cycles: u64,
    pub start_pc: u32,
    pub end_pc: u32,
    pub exit: u32,
    pub pre_state: [u32; 8],
    pub post_state[8]: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_air_step {
    pub op: u32,
    pub a: u32,
    pub b: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_air_info {
    pub n_steps: u64,
    pub n_ops: u64,
    pub n_constraints: u32,
    pub max_degree: u32,
    pub log_quotient_degree: u32,
    pub n_fp_slots: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_p3_table {
    pub trace: *const u32,
    pub log_height: u32,
    pub width: u32,
    pub air: *const rk_air,
    pub public_values: *const u32,
    pub n_public: u32,
    pub on_device: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_p3_shard {
    pub tables: *const rk_p3_table,
    pub n_tables: u32,
    pub init_words: *const u32,
    pub n_init: usize,
    pub h_proof: *mut u32,
    pub capacity_words: usize,
    pub proof_words: usize,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_p3_session_opts {
    pub device: c_int,
    pub batch: c_int,
    pub verify: c_int,
    pub devices: *const c_int,
    pub n_devices: c_int,
    pub params: *const rk_params,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_p3_timing {
    pub lde: f32,
    pub commit: f32,
    pub quotient: f32,
    pub open: f32,
    pub fri: f32,
    pub query: f32,
    pub total: f32,
    pub perm: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_timing {
    pub ntt: f32,
    pub hash: f32,
    pub deep: f32,
    pub fri: f32,
    pub query: f32,
    pub total: f32,
    pub circuit: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rk_kernel_stat {
    pub launches: u64,
    pub ms: f64,
    pub bytes: f64,
}

#[link(name = "raiko_hip")]
extern "C" {
    pub fn rk_abi_version() -> c_int;
    pub fn rk_strerror(status: c_int) -> *const c_char;
    pub fn rk_last_error(ctx: *mut rk_ctx) -> *const c_char;
    pub fn rk_device_count(count: *mut c_int) -> c_int;
    pub fn rk_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut rk_ctx) -> c_int;
    pub fn rk_ctx_destroy(ctx: *mut rk_ctx) -> c_int;
    pub fn rk_sync(ctx: *mut rk_ctx) -> c_int;
    pub fn rk_alloc(ctx: *mut rk_ctx, bytes: usize, d_ptr: *mut *mut c_void) -> c_int;
    pub fn rk_free(ctx: *mut rk_ctx, d_ptr: *mut c_void) -> c_int;
    pub fn rk_h2d(ctx: *mut rk_ctx, d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> c_int;
    pub fn rk_d2h(ctx: *mut rk_ctx, h_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn rk_set_poseidon2_params(ctx: *mut rk_ctx, rc_ext: *const u32, rc_int: *const u32, diag: *const u32) -> c_int;
    pub fn rk_params_preset(out: *mut rk_params, preset: c_int) -> c_int;
    pub fn rk_set_params(ctx: *mut rk_ctx, params: *const rk_params) -> c_int;
    pub fn rk_get_params(ctx: *mut rk_ctx, out: *mut rk_params) -> c_int;
    pub fn rk_batch_interpolate_ntt(ctx: *mut rk_ctx, d_io: *mut u32, size: usize, count: usize) -> c_int;
    pub fn rk_batch_evaluate_ntt(ctx: *mut rk_ctx, d_io: *mut u32, size: usize, count: usize, expand_bits: u32) -> c_int;
    pub fn rk_zk_shift(ctx: *mut rk_ctx, d_io: *mut u32, size: usize, count: usize) -> c_int;
    pub fn rk_batch_expand_into_evaluate_ntt(ctx: *mut rk_ctx, d_out: *mut u32, d_in: *const u32, in_size: usize, count: usize, expand_bits: u32) -> c_int;
    pub fn rk_batch_bit_reverse(ctx: *mut rk_ctx, d_io: *mut u32, size: usize, count: usize) -> c_int;
    pub fn rk_hash_rows(ctx: *mut rk_ctx, d_out_digests: *mut u32, d_matrix: *const u32, rows: usize, cols: usize) -> c_int;
    pub fn rk_hash_fold(ctx: *mut rk_ctx, d_nodes: *mut u32, input_size: usize, output_size: usize) -> c_int;
    pub fn rk_batch_evaluate_any(ctx: *mut rk_ctx, d_coeffs: *const u32, poly_count: usize, size: usize, h_which: *const u32, h_xs: *const u32, eval_count: usize, h_out: *mut u32) -> c_int;
    pub fn rk_mix_poly_coeffs(ctx: *mut rk_ctx, d_out_ext: *mut u32, mix_start: *const u32, mix: *const u32, d_in: *const u32, h_combos: *const u32, input_size: usize, count: usize) -> c_int;
    pub fn rk_eltwise_add_elem(ctx: *mut rk_ctx, d_out: *mut u32, d_a: *const u32, d_b: *const u32, n: usize) -> c_int;
    pub fn rk_eltwise_sum_extelem(ctx: *mut rk_ctx, d_out: *mut u32, d_in_ext: *const u32, count: usize, to_add: usize) -> c_int;
    pub fn rk_eltwise_copy_elem(ctx: *mut rk_ctx, d_out: *mut u32, d_in: *const u32, n: usize) -> c_int;
    pub fn rk_eltwise_zeroize_elem(ctx: *mut rk_ctx, d_io: *mut u32, n: usize) -> c_int;
    pub fn rk_fri_fold(ctx: *mut rk_ctx, d_out: *mut u32, d_in: *const u32, out_count: usize, mix: *const u32) -> c_int;
    pub fn rk_fri_fold_evals(ctx: *mut rk_ctx, d_out_ext: *mut u32, d_in_ext: *const u32, n_out: usize, beta: *const u32) -> c_int;
    pub fn rk_gather_sample(ctx: *mut rk_ctx, d_dst: *mut u32, d_src: *const u32, idx: usize, size: usize, stride: usize) -> c_int;
    pub fn rk_prefix_products(ctx: *mut rk_ctx, d_io_ext: *mut u32, count: usize) -> c_int;
    pub fn rk_scatter(ctx: *mut rk_ctx, d_into: *mut u32, into_words: usize, h_index: *const u32, n_cycles: usize, h_offsets: *const u32, h_values: *const u32) -> c_int;
    pub fn rk_pow_grind(ctx: *mut rk_ctx, sponge_cells: *const u32, bits: u32, nonce: *mut u32) -> c_int;
    pub fn rk_merkle_build(ctx: *mut rk_ctx, d_nodes: *mut u32, d_matrix: *const u32, rows: usize, cols: usize) -> c_int;
    pub fn rk_mmcs_commit(ctx: *mut rk_ctx, mats: *const rk_matrix, n_mats: u32, d_nodes: *mut u32, h_root: *mut u32) -> c_int;
    pub fn rk_mmcs_open(ctx: *mut rk_ctx, mats: *const rk_matrix, n_mats: u32, d_nodes: *const u32, index: u32, h_rows: *mut u32, h_path: *mut u32) -> c_int;
    pub fn rk_mmcs_verify(params: *const rk_params, heights: *const u32, widths: *const u32, n_mats: u32, index: u32, rows: *const u32, path: *const u32, root: *const u32) -> c_int;
    pub fn rk_pcs_coset_lde_rows(ctx: *mut rk_ctx, d_out: *mut u32, d_in: *const u32, height: usize, width: usize) -> c_int;
    pub fn rk_pcs_eval_at(ctx: *mut rk_ctx, d_out_ext: *mut u32, d_lde: *const u32, lde_height: usize, width: usize, z: *const u32) -> c_int;
    pub fn rk_pcs_eval_at_many(ctx: *mut rk_ctx, d_out_ext: *mut u32, d_lde: *const u32, lde_height: usize, width: usize, n_points: u32, h_points: *const u32) -> c_int;
    pub fn rk_pcs_reduce_openings(ctx: *mut rk_ctx, d_ro_ext: *mut u32, d_lde: *const u32, lde_height: usize, width: usize, n_points: u32, h_points: *const u32, h_opened: *const u32, alpha: *const u32, alpha_offset: u64) -> c_int;
    pub fn rk_pcs_coset_lde_cols(ctx: *mut rk_ctx, d_cols: *mut u32, d_in_rows: *const u32, height: usize, width: usize) -> c_int;
    pub fn rk_pcs_eval_at_many_cols(ctx: *mut rk_ctx, d_out_ext: *mut u32, d_lde_cols: *const u32, lde_height: usize, width: usize, n_points: u32, h_points: *const u32) -> c_int;
    pub fn rk_pcs_reduce_openings_cols(ctx: *mut rk_ctx, d_ro_ext: *mut u32, d_lde_cols: *const u32, lde_height: usize, width: usize, n_points: u32, h_points: *const u32, h_opened: *const u32, alpha: *const u32, alpha_offset: u64) -> c_int;
    pub fn rk_duplex_grind(ctx: *mut rk_ctx, sponge_state: *const u32, input_buffer: *const u32, n_input: u32, bits: u32, witness: *mut u32) -> c_int;
    pub fn rk_poly_divide(ctx: *mut rk_ctx, d_polys_ext: *mut u32, count: usize, z: *const u32, h_rem: *mut u32) -> c_int;
    pub fn rk_program_create(steps: *const rk_poly_step, n_steps: usize, ret: u32, taps: *const rk_taps, out: *mut *mut rk_program) -> c_int;
    pub fn rk_program_destroy(prog: *mut rk_program) -> c_int;
    pub fn rk_program_get_info(prog: *const rk_program, out: *mut rk_program_info) -> c_int;
    pub fn rk_program_eval_check(prog: *const rk_program, view: *const rk_circuit_view, poly_mix: *const u32, d_check: *mut u32) -> c_int;
    pub fn rk_program_compile(prog: *mut rk_program, ctx: *mut rk_ctx) -> c_int;
    pub fn rk_program_source(prog: *const rk_program, out: *mut c_char, capacity: usize, length: *mut usize) -> c_int;
    pub fn rk_program_poly_ext(prog: *const rk_program, ext_w: u32, poly_mix: *const u32, eval_u_ext: *const u32, n_taps: usize, globals: *const u32, n_globals: u32, mix: *const u32, n_mix: u32, out_ext: *mut u32) -> c_int;
    pub fn rk_prove_segment(ctx: *mut rk_ctx, seg: *const rk_segment, h_seal: *mut u32, seal_capacity_words: usize, seal_words: *mut usize) -> c_int;
    pub fn rk_verify_segment(pub_: *const rk_segment, seal: *const u32, seal_words: usize) -> c_int;
    pub fn rk_verify_segment_ex(pub_: *const rk_segment, opts: *const rk_verify_opts, seal: *const u32, seal_words: usize) -> c_int;
    pub fn rk_seal_bound_words(seg: *const rk_segment) -> usize;
    pub fn rk_seal_bound_words_for(seg: *const rk_segment, queries: u32) -> usize;
    pub fn rk_seal_bound_words_params(seg: *const rk_segment, params: *const rk_params) -> usize;
    pub fn rk_prove_session(opts: *const rk_session_opts, segs: *const rk_segment, n: usize, h_seals: *const *mut u32, seal_capacity_words: *const usize, seal_words: *mut usize, failed_index: *mut usize) -> c_int;
    pub fn rk_stream_open(opts: *const rk_session_opts, out: *mut *mut rk_stream) -> c_int;
    pub fn rk_stream_submit(stream: *mut rk_stream, seg: *const rk_segment, h_seal: *mut u32, seal_capacity_words: usize, seal_words: *mut usize) -> c_int;
    pub fn rk_stream_wait(stream: *mut rk_stream, max_pending: usize, finished_prefix: *mut usize) -> c_int;
    pub fn rk_stream_close(stream: *mut rk_stream, failed_index: *mut usize) -> c_int;
    pub fn rk_session_last_error(device: c_int) -> *const c_char;
    pub fn rk_session_last_proven(device: c_int, count: *mut usize) -> c_int;
    pub fn rk_session_release() -> c_int;
    pub fn rk_comm_unique_id(id: *mut u8) -> c_int;
    pub fn rk_comm_create(id: *const u8, rank: c_int, world: c_int, device: c_int, out: *mut *mut rk_comm) -> c_int;
    pub fn rk_comm_destroy(comm: *mut rk_comm) -> c_int;
    pub fn rk_comm_last_error(comm: *mut rk_comm) -> *const c_char;
    pub fn rk_gather_seals(comm: *mut rk_comm, h_local_seals: *const *const u32, local_words: *const usize, n_local: usize, n_total: usize, h_out: *const *mut u32, out_capacity: *const usize, out_words: *mut usize) -> c_int;
    pub fn rk_gather_unpack(all_lens: *const u32, all_payload: *const u32, world: c_int, per_rank: usize, max_len: usize, n_total: usize, h_out: *const *mut u32, out_capacity: *const usize, out_words: *mut usize) -> c_int;
    pub fn rk_exec_elf(elf: *const u8, elf_bytes: usize, opts: *const rk_exec_opts, out: *mut *mut rk_exec) -> c_int;
    pub fn rk_exec_open(elf: *const u8, elf_bytes: usize, opts: *const rk_exec_opts, out: *mut *mut rk_exec) -> c_int;
    pub fn rk_exec_next_segment(ex: *mut rk_exec, more: *mut c_int) -> c_int;
    pub fn rk_exec_summary_get(ex: *const rk_exec, out: *mut rk_exec_summary) -> c_int;
    pub fn rk_exec_segment_get(ex: *const rk_exec, index: u32, out: *mut rk_exec_segment) -> c_int;
    pub fn rk_exec_journal(ex: *const rk_exec, out: *mut u8, capacity: usize, len: *mut usize) -> c_int;
    pub fn rk_exec_profile(ex: *const rk_exec, pcs: *mut u32, cycles: *mut u64, capacity: usize, n: *mut usize) -> c_int;
    pub fn rk_exec_witness(ex: *const rk_exec, index: u32, code: *mut u32, data: *mut u32) -> c_int;
    pub fn rk_exec_lookup_tables(ex: *const rk_exec, index: u32, range_table: *mut u32, program_table: *mut u32, program_rows: *mut usize) -> c_int;
    pub fn rk_exec_witness_device(ctx: *mut rk_ctx, ex: *const rk_exec, index: u32, d_code: *mut u32, d_data: *mut u32) -> c_int;
    pub fn rk_exec_witness_device_rows(ctx: *mut rk_ctx, ex: *const rk_exec, index: u32, d_rows: *mut u32) -> c_int;
    pub fn rk_exec_error(ex: *const rk_exec) -> *const c_char;
    pub fn rk_exec_free(ex: *mut rk_exec) -> c_int;
    pub fn rk_air_create(steps: *const rk_air_step, n_steps: usize, width: u32, n_public: u32, out: *mut *mut rk_air) -> c_int;
    pub fn rk_air_create_lookup(steps: *const rk_air_step, n_steps: usize, width: u32, n_public: u32, interaction_words: *const u32, n_interactions: u32, n_words: usize, ext_w: u32, out: *mut *mut rk_air) -> c_int;
    pub fn rk_air_get_steps(air: *const rk_air, out: *mut rk_air_step, capacity: usize, n_steps: *mut usize) -> c_int;
    pub fn rk_air_destroy(air: *mut rk_air) -> c_int;
    pub fn rk_air_get_info(air: *const rk_air, out: *mut rk_air_info) -> c_int;
    pub fn rk_air_compile(air: *mut rk_air, ctx: *mut rk_ctx) -> c_int;
    pub fn rk_p2_chip_width(params: *const rk_params) -> u32;
    pub fn rk_p2_chip_air(params: *const rk_params, bus: u32, out: *mut *mut rk_air) -> c_int;
    pub fn rk_p2_chip_trace(ctx: *mut rk_ctx, d_inputs: *const u32, d_mult: *const u32, n: usize, d_trace: *mut u32) -> c_int;
    pub fn rk_p3_prove(ctx: *mut rk_ctx, tables: *const rk_p3_table, n_tables: u32, init_words: *const u32, n_init: usize, h_proof: *mut u32, capacity_words: usize, proof_words: *mut usize) -> c_int;
    pub fn rk_p3_verify(params: *const rk_params, tables: *const rk_p3_table, n_tables: u32, init_words: *const u32, n_init: usize, proof: *const u32, proof_words: usize) -> c_int;
    pub fn rk_p3_verify_hashes(params: *const rk_params, tables: *const rk_p3_table, n_tables: u32, init_words: *const u32, n_init: usize, proof: *const u32, proof_words: usize, states: *mut u32, capacity_permutations: usize, n_permutations: *mut usize) -> c_int;
    pub fn rk_p3_proof_bound_words(params: *const rk_params, tables: *const rk_p3_table, n_tables: u32) -> usize;
    pub fn rk_p3_prove_shards(opts: *const rk_p3_session_opts, shards: *mut rk_p3_shard, n: usize, failed_index: *mut usize) -> c_int;
    pub fn rk_p3_last_timing(ctx: *mut rk_ctx, out: *mut rk_p3_timing) -> c_int;
    pub fn rk_last_timing(ctx: *mut rk_ctx, out: *mut rk_timing) -> c_int;
    pub fn rk_set_kernel_timing(ctx: *mut rk_ctx, enabled: c_int) -> c_int;
    pub fn rk_kernel_stats(ctx: *mut rk_ctx, kclass: c_int, out: *mut rk_kernel_stat) -> c_int;
    pub fn rk_kernel_class_name(kclass: c_int) -> *const c_char;
    pub fn rk_session_set_kernel_timing(device: c_int, enabled: c_int) -> c_int;
    pub fn rk_session_kernel_stats(device: c_int, kclass: c_int, out: *mut rk_kernel_stat) -> c_int;
}
