"""`HipHal`: the MI355X implementation of risc0's `Hal` operator set, over the C ABI.

Method names and argument meaning follow `risc0_zkp::hal::Hal` (risc0-zkp 1.0.1, the
trait behind `session.prove()` at reference provers/risc0/driver/src/bonsai.rs:271) so
parity tests read like risc0's own HAL tests.  Buffers are device memory; elements are
BabyBear Montgomery residues (uint32).  No CPU fallback exists: construction fails if
libraiko_hip.so or a GPU is missing.
"""
import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .segment import Segment, TapSet


def _u32p(a: np.ndarray):
    return a.ctypes.data_as(_lib.u32p)


class DeviceBuffer:
    """`Hal::Buffer<u32>`: a device allocation owned by a HipHal context."""

    def __init__(self, hal: "HipHal", words: int):
        self.hal = hal
        self.words = int(words)
        ptr = C.c_void_p()
        _lib.check(hal._ctx, hal._lib.rk_alloc(hal._ctx, self.words * 4, C.byref(ptr)))
        self.ptr = ptr.value

    def size(self) -> int:
        return self.words

    def copy_from(self, host: np.ndarray) -> "DeviceBuffer":
        host = np.ascontiguousarray(host, dtype=np.uint32).reshape(-1)
        assert host.size == self.words, (host.size, self.words)
        _lib.check(self.hal._ctx, self.hal._lib.rk_h2d(self.hal._ctx, self.ptr, host.ctypes.data, host.nbytes))
        return self

    def to_host(self) -> np.ndarray:
        out = np.empty(self.words, dtype=np.uint32)
        _lib.check(self.hal._ctx, self.hal._lib.rk_d2h(self.hal._ctx, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr is not None and self.hal._ctx:
            self.hal._lib.rk_free(self.hal._ctx, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(buf) -> int:
    """Accept a DeviceBuffer, a torch CUDA tensor (uint32/int32) or a raw device address."""
    if isinstance(buf, DeviceBuffer):
        return buf.ptr
    if hasattr(buf, "data_ptr"):
        return buf.data_ptr()
    return int(buf)


class HipHal:
    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._lib = _lib.load()
        self._ctx = None
        n = C.c_int(0)
        st = self._lib.rk_device_count(C.byref(n))
        if st != 0 or n.value <= 0:
            raise _lib.HipLibraryError("no AMD GPU visible to HIP: raiko_amd has no CPU fallback")
        ctx = C.c_void_p()
        _lib.check(None, self._lib.rk_ctx_create(device, stream, C.byref(ctx)))
        self._ctx = ctx.value
        self.device = device

    def close(self):
        if self._ctx:
            self._lib.rk_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        _lib.check(self._ctx, st)

    # ---- buffers ----
    def alloc_elem(self, words: int) -> DeviceBuffer:
        return DeviceBuffer(self, words)

    def copy_from_elem(self, host: np.ndarray) -> DeviceBuffer:
        host = np.ascontiguousarray(host, dtype=np.uint32)
        return DeviceBuffer(self, host.size).copy_from(host)

    def sync(self):
        self._ck(self._lib.rk_sync(self._ctx))

    def set_poseidon2_params(self, rc_ext: np.ndarray, rc_int: np.ndarray, diag: np.ndarray):
        """Replace the Poseidon2 instance of this context: 192 external and 21 internal round
        constants and the 24 internal-diagonal entries, Montgomery form (rk_set_poseidon2_params)."""
        a = [np.ascontiguousarray(x, dtype=np.uint32).reshape(-1) for x in (rc_ext, rc_int, diag)]
        if [x.size for x in a] != [192, 21, 24]:
            raise ValueError("expected 192 + 21 + 24 constants")
        self._ck(self._lib.rk_set_poseidon2_params(self._ctx, _u32p(a[0]), _u32p(a[1]), _u32p(a[2])))

    def set_params(self, preset: int = 0, **over):
        """rk_set_params: a preset (_lib.RK_PRESET_RISC0 / RK_PRESET_SP1) with overrides -- ext_w,
        root_2_27, coset_shift (canonical integers), p2_width / p2_m4 / p2_pad_free, p2_rc_ext /
        p2_rc_int / p2_diag (uint32 arrays, Montgomery form), queries, blowup_log2, fri_fold_log2,
        fri_min_degree.  Returns the blob applied (keep it for verify_segment(params=...))."""
        p = make_params(preset, **over)
        self._ck(self._lib.rk_set_params(self._ctx, C.byref(p)))
        self._params = p
        return p

    def get_params(self) -> "_lib.RkParams":
        p = _lib.RkParams()
        self._ck(self._lib.rk_get_params(self._ctx, C.byref(p)))
        return p

    # ---- Hal operators ----
    def batch_interpolate_ntt(self, io, count: int, size: Optional[int] = None):
        size = size if size is not None else io.size() // count
        self._ck(self._lib.rk_batch_interpolate_ntt(self._ctx, _ptr(io), size, count))

    def batch_evaluate_ntt(self, io, count: int, expand_bits: int = 0, size: Optional[int] = None):
        size = size if size is not None else io.size() // count
        self._ck(self._lib.rk_batch_evaluate_ntt(self._ctx, _ptr(io), size, count, expand_bits))

    def zk_shift(self, io, count: int, size: Optional[int] = None):
        size = size if size is not None else io.size() // count
        self._ck(self._lib.rk_zk_shift(self._ctx, _ptr(io), size, count))

    def batch_expand_into_evaluate_ntt(self, out, inp, count: int, expand_bits: int, in_size: Optional[int] = None):
        in_size = in_size if in_size is not None else inp.size() // count
        self._ck(self._lib.rk_batch_expand_into_evaluate_ntt(self._ctx, _ptr(out), _ptr(inp), in_size, count, expand_bits))

    def batch_bit_reverse(self, io, count: int, size: Optional[int] = None):
        size = size if size is not None else io.size() // count
        self._ck(self._lib.rk_batch_bit_reverse(self._ctx, _ptr(io), size, count))

    def hash_rows(self, out, matrix, rows: int, cols: int):
        self._ck(self._lib.rk_hash_rows(self._ctx, _ptr(out), _ptr(matrix), rows, cols))

    def hash_fold(self, io, input_size: int, output_size: int):
        self._ck(self._lib.rk_hash_fold(self._ctx, _ptr(io), input_size, output_size))

    def merkle_build(self, nodes, matrix, rows: int, cols: int):
        self._ck(self._lib.rk_merkle_build(self._ctx, _ptr(nodes), _ptr(matrix), rows, cols))

    def batch_evaluate_any(self, coeffs, poly_count: int, size: int, which: Sequence[int], xs: np.ndarray) -> np.ndarray:
        which = np.ascontiguousarray(which, dtype=np.uint32)
        xs = np.ascontiguousarray(xs, dtype=np.uint32).reshape(-1, 4)
        assert xs.shape[0] == which.size
        out = np.zeros((which.size, 4), dtype=np.uint32)
        self._ck(self._lib.rk_batch_evaluate_any(self._ctx, _ptr(coeffs), poly_count, size, _u32p(which), _u32p(xs),
                                                 which.size, _u32p(out)))
        return out

    def mix_poly_coeffs(self, out, mix_start, mix, inp, combos: Sequence[int], input_size: int, count: int):
        ms = np.ascontiguousarray(mix_start, dtype=np.uint32)
        mx = np.ascontiguousarray(mix, dtype=np.uint32)
        cb = np.ascontiguousarray(combos, dtype=np.uint32)
        assert cb.size == input_size
        self._ck(self._lib.rk_mix_poly_coeffs(self._ctx, _ptr(out), _u32p(ms), _u32p(mx), _ptr(inp), _u32p(cb),
                                              input_size, count))

    def eltwise_add_elem(self, out, a, b, n: int):
        self._ck(self._lib.rk_eltwise_add_elem(self._ctx, _ptr(out), _ptr(a), _ptr(b), n))

    def eltwise_sum_extelem(self, out, inp, count: int, to_add: int):
        self._ck(self._lib.rk_eltwise_sum_extelem(self._ctx, _ptr(out), _ptr(inp), count, to_add))

    def eltwise_copy_elem(self, out, inp, n: int):
        self._ck(self._lib.rk_eltwise_copy_elem(self._ctx, _ptr(out), _ptr(inp), n))

    def eltwise_zeroize_elem(self, io, n: int):
        self._ck(self._lib.rk_eltwise_zeroize_elem(self._ctx, _ptr(io), n))

    def fri_fold(self, out, inp, out_count: int, mix):
        mx = np.ascontiguousarray(mix, dtype=np.uint32)
        self._ck(self._lib.rk_fri_fold(self._ctx, _ptr(out), _ptr(inp), out_count, _u32p(mx)))

    def fri_fold_evals(self, out, inp, n_out: int, beta):
        """Plonky3's arity-2 fold on bit-reversed evaluations (interleaved extension elements)"""
        b = np.ascontiguousarray(beta, dtype=np.uint32)
        self._ck(self._lib.rk_fri_fold_evals(self._ctx, _ptr(out), _ptr(inp), n_out, _u32p(b)))

    # ---- Plonky3 two-adic FRI PCS steps on row-major matrices (SP1's prover: provers/sp1/driver/src/lib.rs:48-57) ----
    def pcs_coset_lde_rows(self, out, inp, height: int, width: int):
        """TwoAdicFriPcs::commit for one matrix: coset LDE (blow-up / shift of the parameter set), bit-reversed rows"""
        self._ck(self._lib.rk_pcs_coset_lde_rows(self._ctx, _ptr(out), _ptr(inp), height, width))

    def pcs_coset_lde_cols(self, out, inp, height: int, width: int):
        """the same LDE left the way the NTT leaves it: `width` columns of (height << blow-up) natural-order evaluations
        (rk_matrix layout 2); `inp` is the row-major trace"""
        self._ck(self._lib.rk_pcs_coset_lde_cols(self._ctx, _ptr(out), _ptr(inp), height, width))

    def pcs_eval_at_many_cols(self, lde_cols, lde_height: int, width: int, points) -> np.ndarray:
        pts = np.ascontiguousarray(points, dtype=np.uint32).reshape(-1, 4)
        out = self.alloc_elem(pts.shape[0] * width * 4)
        self._ck(self._lib.rk_pcs_eval_at_many_cols(self._ctx, _ptr(out), _ptr(lde_cols), lde_height, width, pts.shape[0], _u32p(pts)))
        return out.to_host().reshape(pts.shape[0], width, 4)

    def pcs_reduce_openings_cols(self, ro, lde_cols, lde_height: int, width: int, points, opened, alpha, alpha_offset: int = 0):
        pts = np.ascontiguousarray(points, dtype=np.uint32).reshape(-1, 4)
        ys = np.ascontiguousarray(opened, dtype=np.uint32).reshape(pts.shape[0], width, 4)
        a = np.ascontiguousarray(alpha, dtype=np.uint32)
        self._ck(self._lib.rk_pcs_reduce_openings_cols(self._ctx, _ptr(ro), _ptr(lde_cols), lde_height, width, pts.shape[0], _u32p(pts),
                                                       _u32p(ys), _u32p(a), alpha_offset))

    def pcs_eval_at(self, lde, lde_height: int, width: int, z) -> np.ndarray:
        """the opened values p_c(z) of one committed matrix: (width, 4) words"""
        zz = np.ascontiguousarray(z, dtype=np.uint32)
        out = self.alloc_elem(width * 4)
        self._ck(self._lib.rk_pcs_eval_at(self._ctx, _ptr(out), _ptr(lde), lde_height, width, _u32p(zz)))
        return out.to_host().reshape(width, 4)

    def pcs_eval_at_many(self, lde, lde_height: int, width: int, points) -> np.ndarray:
        """opened values at up to four points in one pass over the low coset: (n_points, width, 4) words"""
        pts = np.ascontiguousarray(points, dtype=np.uint32).reshape(-1, 4)
        out = self.alloc_elem(pts.shape[0] * width * 4)
        self._ck(self._lib.rk_pcs_eval_at_many(self._ctx, _ptr(out), _ptr(lde), lde_height, width, pts.shape[0], _u32p(pts)))
        return out.to_host().reshape(pts.shape[0], width, 4)

    def pcs_reduce_openings(self, ro, lde, lde_height: int, width: int, points, opened, alpha, alpha_offset: int = 0):
        """ro[r] += alpha^(offset + j width) (sum_c alpha^c M[r][c] - sum_c alpha^c opened_j[c]) / (x_r - z_j), all j"""
        pts = np.ascontiguousarray(points, dtype=np.uint32).reshape(-1, 4)
        ys = np.ascontiguousarray(opened, dtype=np.uint32).reshape(pts.shape[0], width, 4)
        a = np.ascontiguousarray(alpha, dtype=np.uint32)
        self._ck(self._lib.rk_pcs_reduce_openings(self._ctx, _ptr(ro), _ptr(lde), lde_height, width, pts.shape[0], _u32p(pts),
                                                  _u32p(ys), _u32p(a), alpha_offset))

    def duplex_grind(self, sponge_state, input_buffer, bits: int) -> int:
        """Plonky3 DuplexChallenger::grind on a challenger in the given state: the smallest witness (canonical integer)"""
        st = np.ascontiguousarray(sponge_state, dtype=np.uint32)
        inp = np.ascontiguousarray(input_buffer, dtype=np.uint32).reshape(-1)
        w = np.zeros(1, dtype=np.uint32)
        pad = inp if inp.size else np.zeros(1, dtype=np.uint32)
        self._ck(self._lib.rk_duplex_grind(self._ctx, _u32p(st), _u32p(pad), inp.size, bits, _u32p(w)))
        return int(w[0])

    def gather_sample(self, dst, src, idx: int, size: int, stride: int):
        self._ck(self._lib.rk_gather_sample(self._ctx, _ptr(dst), _ptr(src), idx, size, stride))

    def poly_divide(self, poly, count: int, z) -> np.ndarray:
        zz = np.ascontiguousarray(z, dtype=np.uint32)
        rem = np.zeros(4, dtype=np.uint32)
        self._ck(self._lib.rk_poly_divide(self._ctx, _ptr(poly), count, _u32p(zz), _u32p(rem)))
        return rem

    # ---- mixed-matrix commitment (Plonky3 MerkleTreeMmcs) ----
    def _c_mats(self, mats):
        arr = (_lib.RkMatrix * len(mats))()
        for i, (buf, height, width, row_major) in enumerate(mats):
            arr[i].d_values = _ptr(buf)
            arr[i].height, arr[i].width, arr[i].row_major = int(height), int(width), int(row_major)   # 0 / 1 / 2 (True = 1)
        return arr

    def mmcs_commit(self, mats):
        """mats: [(device buffer, height, width, layout)] in commit order -> (nodes buffer, root[8]); layout 1 / True =
        row-major, 0 = column-major, 2 = column-major with the committed rows at bit-reversed indices (rk_matrix)"""
        h_max = max(int(m[1]) for m in mats)
        nodes = self.alloc_elem(2 * h_max * 8)
        root = np.zeros(8, dtype=np.uint32)
        self._ck(self._lib.rk_mmcs_commit(self._ctx, self._c_mats(mats), len(mats), _ptr(nodes), _u32p(root)))
        return nodes, root

    def mmcs_open(self, mats, nodes, index: int):
        """-> (rows: the opened row of every matrix concatenated, path: log2(H) x 8 sibling digests)"""
        h_max = max(int(m[1]) for m in mats)
        rows = np.zeros(sum(int(m[2]) for m in mats), dtype=np.uint32)
        path = np.zeros((max(h_max.bit_length() - 1, 0), 8), dtype=np.uint32)
        pp = path if path.size else np.zeros((1, 8), dtype=np.uint32)
        self._ck(self._lib.rk_mmcs_open(self._ctx, self._c_mats(mats), len(mats), _ptr(nodes), int(index), _u32p(rows), _u32p(pp)))
        return rows, path

    # ---- whole segment ----
    def prove_segment(self, seg: Segment, device_inputs=None, consume_inputs: bool = False) -> np.ndarray:
        """Seal (uint32 transcript) of one segment.  `device_inputs` = (groups[3], check) of device
        buffers / tensors to prove from HBM-resident inputs; otherwise the host arrays are uploaded.
        `consume_inputs`: the device buffers may be overwritten (no private copy inside the prover)."""
        c_seg, keep = make_c_segment(seg, device_inputs)
        if consume_inputs and device_inputs is not None:
            c_seg.on_device = 2
        prm = self.get_params()
        cap = int(self._lib.rk_seal_bound_words_params(C.byref(c_seg), C.byref(prm)))
        if cap == 0:  # malformed shape / tap set: nothing is allocated for it
            raise _lib.RkError(_lib.RK_ERR_INVALID, "invalid argument (segment shape or tap set)")
        seal = np.empty(cap, dtype=np.uint32)
        words = C.c_size_t(0)
        self._ck(self._lib.rk_prove_segment(self._ctx, C.byref(c_seg), _u32p(seal), cap, C.byref(words)))
        del keep
        return seal[:words.value].copy()

    def set_kernel_timing(self, enabled: bool):
        self._ck(self._lib.rk_set_kernel_timing(self._ctx, 1 if enabled else 0))

    def kernel_stats(self) -> dict:
        """{kernel class name: {launches, ms, bytes}} since set_kernel_timing(True)"""
        out = {}
        for k in range(_lib.KCLASS_COUNT):
            st = _lib.RkKernelStat()
            self._ck(self._lib.rk_kernel_stats(self._ctx, k, C.byref(st)))
            out[self._lib.rk_kernel_class_name(k).decode()] = {
                "launches": int(st.launches), "ms": float(st.ms), "bytes": float(st.bytes)}
        return out

    def last_timing(self) -> dict:
        t = _lib.RkTiming()
        self._ck(self._lib.rk_last_timing(self._ctx, C.byref(t)))
        return {n: getattr(t, n) for n, _ in t._fields_}


def prove_session(segments, device: int = 0, inflight: int = 3, upload_ahead: int = 2, verify: bool = True,
                  device_inputs=None, devices: Optional[Sequence[int]] = None, poly_ext=None, poseidon2=None, program=None,
                  params=None):
    """Seals of all `segments`, in order, through rk_prove_session: `inflight` proofs in flight on
    each GPU (`devices`: several GPUs of the node share one work queue of segments), host-resident
    traces staged `upload_ahead` segments ahead on a separate stream, every seal verified on a host
    thread of its own (raiko_amd/csrc/session.hip; `poly_ext` / `poseidon2` as in verify_segment).  Raises RkError with
    `.status` (RK_ERR_VERIFY = -7 for a seal that does not verify) and `.segment` = failing index."""
    lib = _lib.load()
    n = len(segments)
    if n == 0:
        return []
    c_segs = (_lib.RkSegment * n)()
    keep = []
    for i, seg in enumerate(segments):
        c, k = make_c_segment(seg, device_inputs[i] if device_inputs else None)
        C.memmove(C.byref(c_segs[i]), C.byref(c), C.sizeof(_lib.RkSegment))
        keep.append((c, k))
    caps = (C.c_size_t * n)()
    words = (C.c_size_t * n)()
    ptrs = (_lib.u32p * n)()
    seals = []
    for i in range(n):
        caps[i] = int(lib.rk_seal_bound_words_params(C.byref(c_segs[i]), C.byref(params))) if params is not None else \
            int(lib.rk_seal_bound_words(C.byref(c_segs[i])))
        if caps[i] == 0:
            e = _lib.RkError(_lib.RK_ERR_INVALID, "invalid argument (segment %d: shape or tap set)" % i)
            e.segment = i
            raise e
        buf = np.empty(caps[i], dtype=np.uint32)
        seals.append(buf)
        ptrs[i] = _u32p(buf)
    opts = _lib.RkSessionOpts(device=device, inflight=inflight, upload_ahead=upload_ahead, verify=1 if verify else 0)
    if devices is not None:
        dev_arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        opts.devices = dev_arr
        opts.n_devices = len(devices)
        keep.append(dev_arr)
    if poly_ext is not None or poseidon2 is not None or program is not None:
        vopts, vkeep = make_verify_opts(poly_ext, poseidon2, program=program)
        opts.verify_opts = C.pointer(vopts)
        keep.append((vopts, vkeep))
    if params is not None:  # an rk_params blob (make_params): the session's proofs and their verification use it
        opts.params = C.pointer(params)
        keep.append(params)
    failed = C.c_size_t(0)
    st = lib.rk_prove_session(C.byref(opts), c_segs, n, ptrs, caps, words, C.byref(failed))
    del keep
    if st != 0:
        detail = lib.rk_session_last_error(device).decode()
        which = int(failed.value) if failed.value != C.c_size_t(-1).value else -1
        e = _lib.RkError(st, "%s%s%s" % (lib.rk_strerror(st).decode(), ": " + detail if detail else "",
                                         " (segment %d)" % which if which >= 0 else ""))
        e.segment = which
        raise e
    return [seals[i][: words[i]].copy() for i in range(n)]


class SessionStream:
    """rk_stream_*: a session whose segments arrive while it runs.  submit() hands one segment over (its arrays
    are kept alive here), close() waits for everything and returns the seals in submission order."""

    def __init__(self, device: int = 0, inflight: int = 3, upload_ahead: int = 2, verify: bool = True, program=None,
                 poly_ext=None, params=None, devices: Optional[Sequence[int]] = None):
        self._lib = _lib.load()
        self._keep = []
        self._bufs = []
        self._words = []
        self._device = device
        opts = _lib.RkSessionOpts(device=device, inflight=inflight, upload_ahead=upload_ahead, verify=1 if verify else 0)
        if devices is not None:
            dev_arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            opts.devices = dev_arr
            opts.n_devices = len(devices)
        if poly_ext is not None or program is not None:
            vopts, vkeep = make_verify_opts(poly_ext, program=program)
            opts.verify_opts = C.pointer(vopts)
            self._keep.append((vopts, vkeep))
        self._params = params
        if params is not None:
            opts.params = C.pointer(params)
        h = C.c_void_p()
        _lib.check(None, self._lib.rk_stream_open(C.byref(opts), C.byref(h)))
        self._h = h

    def submit(self, seg: Segment, device_inputs=None):
        """`device_inputs` = (groups[3], check) of device buffers: the segment's columns are already in HBM"""
        c, k = make_c_segment(seg, device_inputs)
        cap = int(self._lib.rk_seal_bound_words_params(C.byref(c), C.byref(self._params))) if self._params is not None else \
            int(self._lib.rk_seal_bound_words(C.byref(c)))
        if cap == 0:
            raise _lib.RkError(_lib.RK_ERR_INVALID, "invalid argument (segment shape or tap set)")
        buf = np.empty(cap, dtype=np.uint32)
        words = C.c_size_t(0)
        self._keep.append((c, k, seg))
        self._bufs.append(buf)
        self._words.append(words)
        _lib.check(None, self._lib.rk_stream_submit(self._h, C.byref(c), _u32p(buf), cap, C.byref(words)))

    def wait(self, max_pending: int) -> int:
        """rk_stream_wait: block until at most `max_pending` submitted segments are unfinished; returns the length of
        the finished prefix (those segments' inputs may be freed).  Raises RkError once a segment has failed."""
        prefix = C.c_size_t(0)
        st = self._lib.rk_stream_wait(self._h, max_pending, C.byref(prefix))
        if st != 0:
            raise _lib.RkError(st, self._lib.rk_strerror(st).decode())
        return int(prefix.value)

    def close(self):
        failed = C.c_size_t(0)
        st = self._lib.rk_stream_close(self._h, C.byref(failed))
        self._h = None
        if st != 0:
            which = int(failed.value) if failed.value != C.c_size_t(-1).value else -1
            detail = self._lib.rk_session_last_error(self._device).decode()
            e = _lib.RkError(st, "%s%s%s" % (self._lib.rk_strerror(st).decode(), ": " + detail if detail else "",
                                             " (segment %d)" % which if which >= 0 else ""))
            e.segment = which
            raise e
        return [b[: w.value].copy() for b, w in zip(self._bufs, self._words)]


_params_keep = []


def make_params(preset: int = 0, **over) -> "_lib.RkParams":
    """an rk_params blob: preset + overrides (see HipHal.set_params); arrays are kept alive by the module"""
    lib = _lib.load()
    p = _lib.RkParams()
    _lib.check(None, lib.rk_params_preset(C.byref(p), preset))
    for k, v in over.items():
        if k in ("p2_rc_ext", "p2_rc_int", "p2_diag"):
            a = np.ascontiguousarray(v, dtype=np.uint32).reshape(-1)
            _params_keep.append(a)
            setattr(p, k, _u32p(a))
        else:
            if not hasattr(p, k):
                raise ValueError("unknown parameter %r" % k)
            setattr(p, k, int(v))
    return p


def session_last_proven(device: int) -> int:
    """segments `device` proved in the last session it took part in"""
    n = C.c_size_t(0)
    _lib.check(None, _lib.load().rk_session_last_proven(device, C.byref(n)))
    return int(n.value)


def session_release():
    """rk_session_release: drop the per-device prover contexts and staging rings the session entry points keep"""
    _lib.check(None, _lib.load().rk_session_release())


def session_set_kernel_timing(device: int, enabled: bool):
    """hipEvent brackets around every launch class of rk_prove_session's contexts of `device`"""
    lib = _lib.load()
    _lib.check(None, lib.rk_session_set_kernel_timing(device, 1 if enabled else 0))


def session_kernel_stats(device: int) -> dict:
    """{kernel class name: {launches, ms, bytes}} summed over the session contexts of `device`"""
    lib = _lib.load()
    out = {}
    for k in range(_lib.KCLASS_COUNT):
        st = _lib.RkKernelStat()
        _lib.check(None, lib.rk_session_kernel_stats(device, k, C.byref(st)))
        out[lib.rk_kernel_class_name(k).decode()] = {"launches": int(st.launches), "ms": float(st.ms), "bytes": float(st.bytes)}
    return out


def mmcs_verify(heights, widths, index: int, rows, path, root, params=None) -> int:
    """Mmcs::verify_batch on the host: 0 = accepted, 1 = rejected (params: an rk_params blob, None = risc0's)"""
    lib = _lib.load()
    h = np.ascontiguousarray(heights, dtype=np.uint32)
    w = np.ascontiguousarray(widths, dtype=np.uint32)
    r = np.ascontiguousarray(rows, dtype=np.uint32)
    p = np.ascontiguousarray(path, dtype=np.uint32).reshape(-1)
    if p.size == 0:
        p = np.zeros(8, dtype=np.uint32)
    rt = np.ascontiguousarray(root, dtype=np.uint32)
    return int(lib.rk_mmcs_verify(C.byref(params) if params is not None else None, _u32p(h), _u32p(w), h.size, int(index), _u32p(r),
                                  _u32p(p), _u32p(rt)))


def make_verify_opts(poly_ext=None, poseidon2=None, params=None, program=None):
    """rk_verify_opts: `poly_ext` = a _lib.POLY_EXT_FN (the circuit's constraint polynomial: the
    verifier then checks the constraint identity); `poseidon2` = (rc_ext[192], rc_int[21], diag[24])
    the seal was produced under (default: the compiled-in instance).  Returns (opts, keepalive)."""
    o = _lib.RkVerifyOpts()
    keep = []
    if poseidon2 is not None:
        a = [np.ascontiguousarray(x, dtype=np.uint32).reshape(-1) for x in poseidon2]
        if [x.size for x in a] != [192, 21, 24]:
            raise ValueError("expected 192 + 21 + 24 constants")
        o.p2_rc_ext, o.p2_rc_int, o.p2_diag = _u32p(a[0]), _u32p(a[1]), _u32p(a[2])
        keep.append(a)
    if poly_ext is not None:
        o.poly_ext = poly_ext
        keep.append(poly_ext)
    if program is not None:  # a circuit_program.Program: the constraint identity from the step list
        o.program = program.handle
        keep.append(program)
    if params is not None:  # the whole blob the seal was produced under (overrides `poseidon2`)
        o.params = C.pointer(params)
        keep.append(params)
    return o, keep


def verify_segment(seg: Segment, seal: np.ndarray, poly_ext=None, poseidon2=None, params=None, program=None) -> int:
    """Host-side check of a seal against the public data of `seg` (no GPU needed): 0 = valid,
    positive = reason code of the first failed check (raiko_amd/csrc/verify.hip)."""
    lib = _lib.load()
    c = _lib.RkSegment()
    keep = []
    c.po2 = seg.po2
    fill_c_taps(c.taps, seg.taps, keep)
    gl = np.ascontiguousarray(seg.globals_, dtype=np.uint32)
    c.globals = _u32p(gl)
    c.n_globals = gl.size
    c.n_accum_mix = seg.n_accum_mix
    for i in range(16):
        c.proof_system_info[i] = seg.proof_system_info[i]
        c.circuit_info[i] = seg.circuit_info[i]
    s = np.ascontiguousarray(seal, dtype=np.uint32)
    if poly_ext is None and poseidon2 is None and params is None and program is None:
        return int(lib.rk_verify_segment(C.byref(c), _u32p(s), s.size))
    opts, k2 = make_verify_opts(poly_ext, poseidon2, params, program)
    rc = int(lib.rk_verify_segment_ex(C.byref(c), C.byref(opts), _u32p(s), s.size))
    del k2
    return rc


def fill_c_taps(c_taps, taps: TapSet, keep: list):
    arrs = {}
    for name in ("reg_group", "reg_offset", "reg_combo", "combo_off", "combo_backs"):
        a = np.ascontiguousarray(getattr(taps, name), dtype=np.uint32)
        keep.append(a)
        arrs[name] = a
        setattr(c_taps, name, _u32p(a))
    for g in range(3):
        c_taps.group_size[g] = int(taps.group_size[g])
    c_taps.n_regs = taps.n_regs
    c_taps.n_combos = taps.n_combos


def hooks_address(seg: Segment) -> int:
    h = getattr(seg, "hooks", None)
    base = 0 if h is None else int(h() if callable(h) else h)
    prog = getattr(seg, "program", None)
    if prog is not None:  # eval_check from the step program, accumulate from `hooks`
        return prog.hooks(base)
    return base


def make_c_segment(seg: Segment, device_inputs=None):
    keep = []
    c = _lib.RkSegment()
    c.po2 = seg.po2
    fill_c_taps(c.taps, seg.taps, keep)
    c.hooks = hooks_address(seg) or None
    if device_inputs is not None:
        groups, check = device_inputs
        c.on_device = 1
        for g in range(3):
            c.group[g] = _ptr(groups[g]) if groups[g] is not None else None
        c.check = _ptr(check) if check is not None else None
        keep.append(device_inputs)
    else:
        c.on_device = 0
        for g in range(3):
            if seg.groups[g] is None:  # produced by the accumulate hook
                continue
            a = np.ascontiguousarray(seg.groups[g], dtype=np.uint32)
            assert a.shape == (seg.taps.group_size[g], seg.rows)
            keep.append(a)
            c.group[g] = a.ctypes.data
        if seg.check is not None:
            chk = np.ascontiguousarray(seg.check, dtype=np.uint32)
            assert chk.shape[0] == 4 and chk.shape[1] % seg.rows == 0   # 4 x (rows << blowup_log2)
            keep.append(chk)
            c.check = chk.ctypes.data
    gl = np.ascontiguousarray(seg.globals_, dtype=np.uint32)
    keep.append(gl)
    c.globals = _u32p(gl)
    c.n_globals = gl.size
    c.n_accum_mix = seg.n_accum_mix
    assert len(seg.proof_system_info) == 16 and len(seg.circuit_info) == 16
    for i in range(16):
        c.proof_system_info[i] = seg.proof_system_info[i]
        c.circuit_info[i] = seg.circuit_info[i]
    return c, keep
