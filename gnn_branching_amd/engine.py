"""Host plumbing between the reference-shaped Python surface and the C-ABI (include/gnnb.h).

``ScorerEngine`` owns one gnnb handle: the packed GNN weights, the bound verified
network and a workspace per batch size.  Tensors are only carriers of device
memory here (``data_ptr()``); all arithmetic of the hot path runs in libgnnb.so.
"""
import ctypes as C
import time

from array import array as _array

import numpy as np
import torch
from torch import nn

from . import _lib
from .plnn.modules import Flatten


def _is_flatten(layer):
    # the caller may hand over the reference's own plnn.modules.Flatten instances
    return isinstance(layer, Flatten) or type(layer).__name__ == "Flatten"


def state_blob(state_dict):
    """The checkpoint's 52 tensors concatenated in state-dict order (host fp32 array)."""
    parts = [np.asarray(v.detach().cpu().float().numpy() if torch.is_tensor(v) else v, dtype=np.float32).reshape(-1)
             for v in state_dict.values()]
    return np.ascontiguousarray(np.concatenate(parts))


# (out, in) of the 26 Linear layers in state-dict order (graph_conv.py:26-74, :428-437)
GNN_LINEARS = [(64, 3), (64, 64), (64, 2), (64, 64), (64, 128), (64, 64), (64, 7), (64, 64), (64, 128), (64, 64), (64, 128),
               (64, 64), (64, 4), (64, 128), (64, 64), (64, 7), (64, 64), (64, 64), (64, 192), (64, 64), (64, 128), (64, 64),
               (64, 128), (64, 64), (64, 64), (1, 64)]
GNN_BLOB_FLOATS = 117825        # the 52 tensors of GraphNet(2, 64) (graph_conv.py:20-76, :281-305, :428-437)


class BabsrResult:
    """Device outputs of one batched BaBSR scoring: `score` and `intercept_tb` of kw_score_conv.py:86, :103, padded
    (B, R) over the concatenated ReLU layers (already multiplied by the mask)."""
    __slots__ = ("scores", "intercepts", "masks", "relu_sizes")

    def __init__(self, scores, intercepts, masks, relu_sizes):
        self.scores, self.intercepts, self.masks, self.relu_sizes = scores, intercepts, masks, relu_sizes

    def per_layer(self, b):
        """(score list, intercept list, mask list) of subproblem b, one 1-D tensor per ReLU layer."""
        return tuple(list(torch.split(t[b], self.relu_sizes)) for t in (self.scores, self.intercepts, self.masks))


def _or_reduce(status):
    v = 0
    for x in status.cpu().tolist():
        v |= int(x)
    return v


def _raise_for_status(st):
    """status word of gnnb_forward: bit 0 = an embedding was NaN (the reference enters pdb there, graph_conv.py:184-186, :339-341);
    bit 1 = a wait inside a kernel (k_gather_update_q's LDS ring, or k_top's workgroup split waiting for its partner workgroups)
    ran into its iteration cap: a protocol bug, a wedged GPU, or -- for k_top -- partner workgroups kept off the chip by other work
    (handle option "top_split" = 1 turns the split off); results are invalid.  bit 2 = gnnb_scatter_amb_records refused a record image
    (not packed for this binding / batch size, or a record outside its arrays)."""
    if st & 2:
        msg = ("a wait inside a kernel (k_gather_update_q ring or k_top workgroup split) hit its iteration cap (status bit 1); "
               "results are invalid; the handle option top_split=1 disables the k_top split")
        print(f"[gnn_branching_amd] {msg}", flush=True)
        raise RuntimeError(msg)
    if st & 4:
        msg = "gnnb_scatter_amb_records refused a record image (status bit 2): packed for another network or batch size, or corrupt"
        print(f"[gnn_branching_amd] {msg}", flush=True)
        raise RuntimeError(msg)
    if st & 1:
        msg = "mu contains nan"
        print(f"[gnn_branching_amd] {msg}", flush=True)
        raise FloatingPointError(msg)


class ForwardResult:
    """Device outputs of one batched forward.  `ready` (BatchPipeline only): the event behind the forward on the side stream it ran on."""
    __slots__ = ("scores", "decisions", "status", "masks", "ready")

    def __init__(self, scores, decisions, status, masks, ready=None):
        self.scores, self.decisions, self.status, self.masks, self.ready = scores, decisions, status, masks, ready

    def wait(self):
        """Order the caller's current stream behind the forward (a no-op for a forward that ran on that stream)."""
        if self.ready is not None:
            cur = torch.cuda.current_stream(self.scores.device)
            cur.wait_event(self.ready)
            for t in (self.scores, self.decisions, self.status):
                t.record_stream(cur)          # (allocated on the side stream: keep the allocator from recycling them under the caller)
        return self

    def check(self):
        """Synchronises.  Raises like the reference would stop (it enters pdb on NaN embeddings,
        graph_conv.py:184-186, :339-341)."""
        if self.ready is not None:
            self.ready.synchronize()
            cur = torch.cuda.current_stream(self.scores.device)
            for t in (self.scores, self.decisions, self.status):
                t.record_stream(cur)          # (as wait(): allocated on the side stream, from here on used by the caller's)
        _raise_for_status(_or_reduce(self.status))
        return self

    def ragged(self):
        """list of B 1-D tensors: the scores of the ambiguous ReLUs of each sample (graph_conv.py:470)."""
        return [self.scores[b][self.masks[b] != 0] for b in range(self.scores.shape[0])]


class HostFedPipeline:
    """Cross-batch double buffering for batches that arrive as HOST tensors (the reference pays its host->device copies inside the
    call, graph_score.py:26-30; SURVEY 8(d): "H2D reported separately").

    ``submit(*forward_args)`` enqueues the copies of batch i + 1 on a COPY stream while the forward of batch i runs on the caller's
    stream, and returns that batch's ForwardResult without synchronising: `depth` (2) sets of device input buffers, each guarded by
    two events -- the copy stream waits for the forward that last read a set before overwriting it, the compute stream waits for the
    set's copies before its forward.  Host tensors are copied straight from where they are, in pieces of 2 MB (see `submit`): from
    pinned memory the copies are asynchronous DMA that hides under the running forward; from pageable memory the runtime stages them
    (the host blocks per piece, the GPU still overlaps them with the previous forward).
    ``compact`` (default): of ``dual_vars`` and ``primals`` the forward reads only the entries of AMBIGUOUS nodes (and primals[-1]), so those
    tensors do not cross the link whole: ``gnnb_pack_amb_records`` gathers, on a few host threads, {index, dual[:, 1], dual[:, 2],
    primal_pre, primal_post} of the nodes with lb < 0 < ub into a pinned image (base B = 256: 1.2 MB instead of 17 MB), the image is
    copied, and one launch (``gnnb_scatter_amb_records``) writes the records into the slot's full-size device tensors in front of the
    forward.  36.5 -> 21 MB per base batch: the copies hide under the forward again.
    Scores are bit-identical to ``engine.forward`` on device-resident inputs (tests/test_gpu_hostfed.py)."""

    PIECE = 1 << 19            # floats per copy (2 MB)
    SMALL = 1 << 20            # bytes: tensors below it are staged together

    def __init__(self, engine, depth=2, compact=True):
        self.eng, self.depth = engine, max(2, int(depth))
        self.compact = bool(compact)
        self.link_bytes = 0
        self.copy_stream = torch.cuda.Stream(device=engine.device)
        self.slots = [None] * self.depth          # per slot: dict(key=shape signature, dev=[tensors], pin=[tensors or None], ev_copy, ev_done)
        self.i = 0

    @staticmethod
    def _flat_inputs(lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, masks):
        ts = list(lower_bounds_all) + list(upper_bounds_all) + list(dual_vars) + list(primals) + [primal_inputs, masks]
        out = []
        for t in ts:
            if not torch.is_tensor(t):
                t = torch.tensor(t, dtype=torch.float32)
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(torch.float32).contiguous()
            out.append(t)
        return out

    @staticmethod
    def _check_sizes(eng, host, nb, nd, npr):
        """ValueError (not a segfault in the C packer) for tensors that do not hold what the bound network needs."""
        B, sizes = int(host[0].shape[0]), eng.sizes
        if nb != len(sizes) or nd != len(sizes) - 2:
            raise ValueError(f"{nb} bound tensors / {nd} dual tensors, layer graph has {len(sizes)} layers")
        for k in range(nb):
            for t, what in ((host[k], "lower"), (host[nb + k], "upper")):
                if t.numel() != B * sizes[k]:
                    raise ValueError(f"{what} bounds of graph layer {k}: {tuple(t.shape)} does not hold {B}x{sizes[k]} values")
        for k in range(nd):
            if host[2 * nb + k].numel() != B * sizes[k + 1] * 3:
                raise ValueError(f"dual_vars[{k}] has {tuple(host[2 * nb + k].shape)}, expected ({B * sizes[k + 1]}, 3)")
        eng._check_primals(eng._net_keepalive, host[2 * nb + nd:2 * nb + nd + npr], B)
        if host[-1].numel() != B * eng.R:
            raise ValueError(f"masks has {tuple(host[-1].shape)}, expected ({B}, {eng.R})")
        if host[-2].numel() != B * sizes[0]:
            raise ValueError("primal_inputs has the wrong size")

    def submit(self, lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, layers, masks):
        eng = self.eng
        host = self._flat_inputs(lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, masks)
        nb, nd, npr = len(lower_bounds_all), len(dual_vars), len(primals)
        compact = self.compact and all(t.device.type == "cpu" for t in host)
        skip = set(range(2 * nb, 2 * nb + nd + npr)) if compact else set()      # dual_vars / primals: records instead of whole tensors
        key = (compact,) + tuple((tuple(t.shape)) for t in host)
        k = self.i % self.depth
        self.i += 1
        sl = self.slots[k]
        if compact:
            # the packer walks raw host pointers with the sizes of whatever network the HANDLE is bound to: bind (cached by key) on every
            # submit -- another pipeline or eng.forward on the shared engine may have rebound it since this slot was shaped -- and check every
            # element count against that binding before the C side sees a pointer
            eng.bind(layers["fixed_layers"], tuple(host[0].shape[1:]))
            self._check_sizes(eng, host, nb, nd, npr)
        with torch.cuda.device(eng.device):
            cur = torch.cuda.current_stream()
            if sl is None or sl["key"] != key:
                if sl is not None:
                    sl["ev_done"].synchronize()
                # tensors below SMALL bytes share ONE pinned staging block and ONE device block (a copy of a few KB costs ~10 us of the copy
                # queue's time each, tools/hostfed_probe.py: a batch has a dozen of them); the big ones get their own device tensors
                small = [j for j, t in enumerate(host) if t.numel() * 4 < self.SMALL and j not in skip]
                offs, tot = {}, 0
                for j in small:
                    offs[j] = tot
                    tot += (host[j].numel() + 63) & ~63
                dev_small = torch.empty(max(tot, 1), dtype=torch.float32, device=eng.device)
                pin_small = torch.empty(max(tot, 1), dtype=torch.float32, pin_memory=True)
                dev = [dev_small[offs[j]:offs[j] + t.numel()].view(t.shape) if j in offs else torch.empty(t.shape, dtype=torch.float32, device=eng.device)
                       for j, t in enumerate(host)]
                sl = {"key": key, "dev": dev, "offs": offs, "dev_small": dev_small, "pin_small": pin_small, "pin_np": pin_small.numpy(),
                      "ev_copy": torch.cuda.Event(), "ev_done": torch.cuda.Event(), "used": False}
                if compact:
                    for j in skip:
                        dev[j].zero_()                # (entries of nodes that are never ambiguous are never read: zero, not garbage)
                    cap = int(eng.lib.gnnb_amb_records_bytes(eng.h, int(host[0].shape[0])))
                    sl["img_cap"] = cap
                    sl["pin_img"] = torch.empty(cap // 4, dtype=torch.int32, pin_memory=True)
                    sl["dev_img"] = torch.empty(cap // 4, dtype=torch.int32, device=eng.device)
                self.slots[k] = sl
            if sl["used"]:
                sl["ev_copy"].synchronize()          # the staging block of this set is free again (its last copies have left the host)
            self.copy_stream.wait_event(sl["ev_done"]) if sl["used"] else None      # the forward that last read this set has finished
            used_words = 0
            if compact:
                B = int(host[0].shape[0])
                tabs = [(C.c_void_p * n)(*[t.data_ptr() for t in g]) for n, g in
                        ((nb, host[:nb]), (nb, host[nb:2 * nb]), (nd, host[2 * nb:2 * nb + nd]), (npr, host[2 * nb + nd:2 * nb + nd + npr]))]
                hb = _lib.Batch(tabs[0], tabs[1], tabs[2], tabs[3], host[-2].data_ptr(), None, None, host[-1].data_ptr(), nb, nd, npr)
                used = C.c_size_t(0)
                _lib.check(eng.lib.gnnb_pack_amb_records(eng.h, C.byref(hb), B, sl["pin_img"].data_ptr(), sl["img_cap"], C.byref(used)),
                           "gnnb_pack_amb_records")
                used_words = (used.value + 3) // 4
            offs, pin_np = sl["offs"], sl["pin_np"]
            for j, o in offs.items():                # (host memcpy of the small tensors into the shared staging block: < 1 MB in all)
                t = host[j]
                if t.device.type == "cpu":
                    pin_np[o:o + t.numel()] = t.reshape(-1).numpy()
            dev_src = [t for t in host if t.device.type != "cpu"]
            if dev_src:
                # sources that already live on the device (or temporaries _flat_inputs made from them) were produced on the caller's
                # stream: the copy stream must not read them before that work is done, nor may the allocator recycle them under it
                self.copy_stream.wait_stream(cur)
                for t in dev_src:
                    t.record_stream(self.copy_stream)
            with torch.cuda.stream(self.copy_stream):
                if offs:
                    if any(host[j].device.type != "cpu" for j in offs):
                        for j, o in offs.items():
                            sl["dev"][j].copy_(host[j], non_blocking=True)
                    else:
                        sl["dev_small"].copy_(sl["pin_small"], non_blocking=True)
                for o in range(0, used_words, self.PIECE):                 # the record image of the ambiguous nodes
                    sl["dev_img"][o:min(o + self.PIECE, used_words)].copy_(sl["pin_img"][o:min(o + self.PIECE, used_words)], non_blocking=True)
                for j, t in enumerate(host):
                    if j in offs or j in skip:
                        continue
                    # pieces of at most 2 MB: measured on MI355X / ROCm 7.2 (tools/hostfed_probe.py), 21 pinned copies of 1.7 MB on a side
                    # stream hide completely under the forward (0.85 ms with or without them), ONE 36.5 MB copy beside the same forward
                    # takes 3.1 ms.  Pageable tensors go through the runtime's own staging (synchronous for the host, still beside the
                    # previous forward on the GPU); packing them into pinned memory here first cost 10 ms per batch.
                    dv, sv = sl["dev"][j].view(-1), t.reshape(-1)
                    for o in range(0, sv.numel(), self.PIECE):
                        dv[o:o + self.PIECE].copy_(sv[o:o + self.PIECE], non_blocking=True)
                sl["ev_copy"].record(self.copy_stream)
            cur.wait_event(sl["ev_copy"])
            self.link_bytes = 4 * (used_words + sum(t.numel() for j, t in enumerate(host) if j not in skip))      # what this submit sent over the link
            d = sl["dev"]
            status = None
            if compact:                              # records -> the slot's full-size dual / primal tensors, one launch in front of the forward
                dptr = (C.c_void_p * nd)(*[t.data_ptr() for t in d[2 * nb:2 * nb + nd]])
                pptr = (C.c_void_p * npr)(*[t.data_ptr() for t in d[2 * nb + nd:2 * nb + nd + npr]])
                status = torch.zeros(2, dtype=torch.int32, device=eng.device)      # [forward, scatter]: the scatter raises bit 2 on a foreign / corrupt image
                _lib.check(eng.lib.gnnb_scatter_amb_records(eng.h, sl["dev_img"].data_ptr(), int(host[0].shape[0]), dptr, nd, pptr, npr,
                                                            status[1:].data_ptr(), C.c_void_p(cur.cuda_stream)), "gnnb_scatter_amb_records")
            res = eng.forward(d[:nb], d[nb:2 * nb], d[2 * nb:2 * nb + nd], d[2 * nb + nd:2 * nb + nd + npr], d[-2], layers, d[-1], status=status)
            # the result outlives the slot: its mask must not be a view of the slot's device buffer, which the submit `depth` calls
            # later overwrites (a held result's ragged() would then be cut with another batch's mask).  Cloned on the compute stream,
            # behind the copies it waited for and in front of ev_done.
            res.masks = res.masks.clone()
            sl["ev_done"].record(cur)
            sl["used"] = True
        return res


class BatchPipeline:
    """`depth` (2) INDEPENDENT batches in flight: one handle (own workspace, own control blocks) and one HIP stream per slot, batches
    dealt to the slots in turn.  A forward is a chain of 11-19 dependent launches; while one batch's kernel drains or its next one
    ramps up, the other batch's kernel fills the CUs: measured per batch on MI355X (tools/two_batches_probe.py) base B=256 0.769 ->
    0.728 ms, deep B=128 0.911 -> 0.785 ms.  Throughput, not latency: a batch takes longer from submit to ready.  Scores are bit-identical to
    ``ScorerEngine.forward`` (tests/test_gpu_pipeline.py).  The handles are created with k_top's workgroup split off (handle option "top_split" = 1):
    with a second batch's kernels on the chip the partner workgroups of a split sample are not guaranteed to be resident together.

    ``submit(*forward_args)`` returns the batch's ForwardResult at once; ``result.wait()`` orders the caller's stream behind it,
    ``result.check()`` synchronises on it; ``synchronize()`` waits for everything submitted."""

    def __init__(self, state_dict, depth=2, T=2, p=64, device=None, options=None):
        self.depth = max(1, int(depth))
        opts = dict(options or {})
        opts["top_split"] = 1             # whatever the caller or the environment says: partner workgroups are not guaranteed to be co-resident here
        self.engines = [ScorerEngine(state_dict, T, p, device, options=opts) for _ in range(self.depth)]
        self.device = self.engines[0].device
        self.streams = self._overlapping_streams(self.device, self.depth)
        self.i = 0

    @staticmethod
    def _overlapping_streams(device, n):
        """n streams whose kernels really run side by side.  HIP multiplexes streams onto a few hardware queues and two streams that share
        one run their kernels in order (measured on MI355X / ROCm 7.2, tools/two_batches_probe.py: of the first nine streams torch hands out
        the pairs (#2, #3) and (#0, #5) serialise, every other pair overlaps) -- so candidates are timed against the streams already
        chosen with two 0.3-ms spin kernels and taken only if the pair finishes in well under twice one kernel's time."""
        if not hasattr(torch.cuda, "_sleep"):                   # (no spin kernel to time with: take the streams as they come)
            return [torch.cuda.Stream(device=device) for _ in range(n)]

        def spin(st):
            with torch.cuda.stream(st):
                torch.cuda._sleep(700000)

        def pair_ms(a, b):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            spin(a)
            spin(b)
            torch.cuda.synchronize(device)
            return 1e3 * (time.perf_counter() - t0)

        with torch.cuda.device(device):
            chosen = [torch.cuda.Stream(device=device)]
            pair_ms(chosen[0], chosen[0])                       # (first launch of the spin kernel)
            serial = min(pair_ms(chosen[0], chosen[0]) for _ in range(2))
            spare = []
            for _ in range(12):
                if len(chosen) == n:
                    break
                cand = torch.cuda.Stream(device=device)
                if all(min(pair_ms(cand, s), pair_ms(cand, s)) < 0.75 * serial for s in chosen):
                    chosen.append(cand)
                else:
                    spare.append(cand)
            chosen += spare[:n - len(chosen)]                    # (no overlapping candidate found: still correct, just no faster)
            while len(chosen) < n:
                chosen.append(torch.cuda.Stream(device=device))
        return chosen

    def submit(self, lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, layers, masks):
        k = self.i % self.depth
        self.i += 1
        eng, st = self.engines[k], self.streams[k]
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            if not cur.query():                                  # the inputs were produced on the caller's stream: wait for it -- unless it is idle:
                st.wait_stream(cur)                              # an event recorded on the default stream orders it against every other stream's
                                                                 # work, and the two batches then run one after the other (0.94 instead of 0.78 ms)
            # Lifetime contract: the caller may drop or overwrite-by-reallocation its inputs as soon as submit returns -- every device
            # input is marked as in use by the side stream, so the caching allocator will not hand its memory out again before the
            # forward has read it.  (Writing INTO an input tensor in place before result.wait() / check() is still a race.)
            for grp in (lower_bounds_all, upper_bounds_all, dual_vars, primals, [primal_inputs, masks]):
                for t in grp:
                    if torch.is_tensor(t) and t.device.type == "cuda":
                        t.record_stream(st)
            with torch.cuda.stream(st):
                res = eng.forward(lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, layers, masks)
                res.ready = torch.cuda.Event()
                res.ready.record(st)
        return res

    def synchronize(self):
        for st in self.streams:
            st.synchronize()


class ScorerEngine:
    def __init__(self, state_dict, T=2, p=64, device=None, options=None):
        """options: {name: int} of handle options (include/gnnb.h gnnb_set_option; _lib.OPTIONS), applied over the ones the
        environment names (_lib.OPTION_ENV: the tests' and bench.py's switches; the library itself reads no environment)."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("gnn_branching_amd needs an AMD GPU (MI355X / gfx950); there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.T, self.p = T, p
        # state_dict None: a handle for the GNN-free entry points only (gnnb_babsr) -- all-zero GNN weights
        blob = state_blob(state_dict) if state_dict is not None else np.zeros(GNN_BLOB_FLOATS, dtype=np.float32)
        self._blob = blob
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gnnb_create(C.byref(h), blob.ctypes.data_as(C.c_void_p), blob.size, T, p), "gnnb_create")
        self.h = h
        self.options = dict(_lib.options_from_env())
        self.options.update(options or {})
        for name, value in self.options.items():
            self.set_option(name, value)
        self._net_key = None
        self._net_keepalive = None
        self.sizes = None
        self.R = 0
        self._ws = {}
        self._prop_cache = {}
        self._prop_host_cache = None
        # a large batch can be cut into `n_streams` contiguous chunks on separate HIP streams; measured slower at every size tried
        # (base B=256: 1 stream 0.84 ms, 2 streams 0.90: every kernel's fixed part is paid twice), so 1 unless a caller sets the attribute
        self.n_streams = 1
        self.min_chunk = 64
        self._streams = []

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.gnnb_destroy(h)
            except Exception:
                pass

    # ---- handle options ---------------------------------------------------------------------
    def set_option(self, name, value):
        """gnnb_set_option: see include/gnnb.h for the table.  "gather" and "dense_lds" must be set before the first bind."""
        _lib.check(self.lib.gnnb_set_option(self.h, name.encode(), int(value)), f"gnnb_set_option({name})")
        self.options[name] = int(value)

    def get_option(self, name):
        v = C.c_int(0)
        _lib.check(self.lib.gnnb_get_option(self.h, name.encode(), C.byref(v)), f"gnnb_get_option({name})")
        return v.value

    # ---- verified network -------------------------------------------------------------------
    def bind(self, fixed_layers, input_shape):
        key = (tuple(id(l) for l in fixed_layers),
               tuple((l.weight.data_ptr(), l.weight._version) for l in fixed_layers if hasattr(l, "weight")),
               tuple(input_shape))
        if key == self._net_key:
            return
        descs = (_lib.LayerDesc * len(fixed_layers))()
        keep = []
        for d, l in zip(descs, fixed_layers):
            if type(l) is nn.Conv2d:
                if l.dilation != (1, 1) or l.groups != 1 or l.stride[0] != l.stride[1] or l.padding[0] != l.padding[1]:
                    raise NotImplementedError(f"unsupported conv geometry: {l}")
                w = np.ascontiguousarray(l.weight.detach().cpu().float().numpy())
                b = np.ascontiguousarray(l.bias.detach().cpu().float().numpy())
                keep += [w, b]
                d.kind, d.c_in, d.c_out = _lib.GNNB_CONV, l.in_channels, l.out_channels
                d.kh, d.kw, d.stride, d.pad = l.kernel_size[0], l.kernel_size[1], l.stride[0], l.padding[0]
                d.weight, d.bias = w.ctypes.data, b.ctypes.data
            elif type(l) is nn.Linear:
                w = np.ascontiguousarray(l.weight.detach().cpu().float().numpy())
                b = np.ascontiguousarray(l.bias.detach().cpu().float().numpy())
                keep += [w, b]
                d.kind, d.n_in, d.n_out = _lib.GNNB_LINEAR, l.in_features, l.out_features
                d.weight, d.bias = w.ctypes.data, b.ctypes.data
            elif type(l) is nn.ReLU:
                d.kind = _lib.GNNB_RELU
            elif _is_flatten(l):
                d.kind = _lib.GNNB_FLATTEN
            else:
                raise NotImplementedError(type(l))        # reference: graph_conv.py:191-192
        c0, h0, w0 = (tuple(input_shape) + (1, 1))[:3]
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gnnb_bind_network(self.h, descs, len(fixed_layers), c0, h0, w0), "gnnb_bind_network")
        ng, R = C.c_int(), C.c_int()
        _lib.check(self.lib.gnnb_graph_info(self.h, C.byref(ng), None, C.byref(R)), "gnnb_graph_info")
        sizes = (C.c_int * ng.value)()
        _lib.check(self.lib.gnnb_graph_info(self.h, C.byref(ng), sizes, C.byref(R)), "gnnb_graph_info")
        self.sizes, self.R = list(sizes), R.value
        self._net_key = key
        self._net_keepalive = list(fixed_layers)
        self._ws.clear()

    # ---- one forward ------------------------------------------------------------------------
    def _dev(self, t):
        if not torch.is_tensor(t):
            t = torch.tensor(t, dtype=torch.float32)       # python lists of LP primals (graph_score.py:30)
        return t.to(device=self.device, dtype=torch.float32, non_blocking=True).contiguous()

    def _prop(self, prop_layers):
        """(B, N_L) weights and (B,) biases of the per-sample property layers (graph_conv.py:196-199)."""
        key = tuple(id(l) for l in prop_layers)
        vkey = tuple((l.weight.data_ptr(), l.weight._version) for l in {id(l): l for l in prop_layers}.values())
        hit = self._prop_cache.get(key)
        if hit is not None and hit[0] == vkey:
            return hit[1], hit[2]
        uniq, index = {}, []
        for l in prop_layers:
            if id(l) not in uniq:
                if l.weight.shape[0] != 1:
                    raise NotImplementedError("the property layer must be Linear(., 1)")   # graph_conv.py:80
                uniq[id(l)] = (len(uniq), l)
            index.append(uniq[id(l)][0])
        w = torch.stack([l.weight.detach()[0].float() for _, l in uniq.values()]).to(self.device)
        b = torch.stack([l.bias.detach()[0].float() for _, l in uniq.values()]).to(self.device)
        idx = torch.tensor(index, device=self.device)
        pw, pb = w[idx].contiguous(), b[idx].contiguous()
        if len(self._prop_cache) > 8:
            self._prop_cache.clear()
        self._prop_cache[key] = (vkey, pw, pb, list(prop_layers))
        return pw, pb

    def workspace(self, B, slot=0):
        ws = self._ws.get((B, slot))
        if ws is None:
            n = self.lib.gnnb_workspace_bytes(self.h, B)
            if n == 0:
                raise RuntimeError("gnnb_workspace_bytes returned 0 (no network bound?)")
            ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            if len(self._ws) > 6:
                self._ws.clear()
            self._ws[(B, slot)] = ws
        return ws

    def _marshal(self, lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, layers, masks):
        """Bind the network, move the arguments of GraphNet.forward to the device and validate their sizes."""
        fixed = layers["fixed_layers"]
        self.bind(fixed, tuple(lower_bounds_all[0].shape[1:]))
        B = int(lower_bounds_all[0].shape[0])
        if len(layers["prop_layers"]) != B:
            raise ValueError(f"{len(layers['prop_layers'])} property layers for a batch of {B}")
        lbs = [self._dev(t) for t in lower_bounds_all]
        ubs = [self._dev(t) for t in upper_bounds_all]
        duals = [self._dev(t) for t in dual_vars]
        prim = [self._dev(t) for t in primals]
        x_lp = self._dev(primal_inputs)
        mask = self._dev(masks)
        ng = len(self.sizes)
        if len(lbs) != ng or len(ubs) != ng:
            raise ValueError(f"{len(lbs)} bound tensors, layer graph has {ng} layers")
        for k, (l, u) in enumerate(zip(lbs, ubs)):
            if l.numel() != B * self.sizes[k] or u.numel() != B * self.sizes[k]:
                raise ValueError(f"bounds of graph layer {k}: {tuple(l.shape)} does not hold {B}x{self.sizes[k]} values")
        for k, d in enumerate(duals):
            if d.numel() != B * self.sizes[k + 1] * 3:
                raise ValueError(f"dual_vars[{k}] has {tuple(d.shape)}, expected ({B * self.sizes[k + 1]}, 3)")
        if mask.numel() != B * self.R:
            raise ValueError(f"masks has {tuple(mask.shape)}, expected ({B}, {self.R})")
        if x_lp.numel() != B * self.sizes[0]:
            raise ValueError("primal_inputs has the wrong size")
        self._check_primals(fixed, prim, B)
        pw, pb = self._prop(layers["prop_layers"])
        self._last_bounds = list(zip(lbs, ubs))           # for mu() (inspection)
        return B, lbs, ubs, duals, prim, x_lp, mask, pw, pb

    def forward(self, lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, layers, masks, status=None):
        """status: optional preallocated device int32 tensor; the forward's status words go to its first element(s), further elements
        (HostFedPipeline: the scatter launch's word) are left to the caller and OR-ed in by ForwardResult.check()."""
        B, lbs, ubs, duals, prim, x_lp, mask, pw, pb = self._marshal(lower_bounds_all, upper_bounds_all, dual_vars, primals,
                                                                     primal_inputs, layers, masks)
        scores = torch.empty(B, self.R, dtype=torch.float32, device=self.device)
        dec = torch.empty(B, 2, dtype=torch.int32, device=self.device)
        nchunk = self.n_streams if (self.n_streams > 1 and B >= self.n_streams * self.min_chunk) else 1
        if status is None:
            status = torch.empty(nchunk, dtype=torch.int32, device=self.device)
        elif status.numel() < nchunk or status.dtype != torch.int32 or status.device != self.device:
            raise ValueError("forward: `status` must be a device int32 tensor with one element per chunk")
        mask2 = mask.view(B, self.R)
        bounds = [(B * c) // nchunk for c in range(nchunk + 1)]

        def launch(c, stream_ptr):
            lo, hi = bounds[c], bounds[c + 1]
            n = hi - lo

            def rows(t):                      # batch-major tensors: rows [lo, hi) of the leading B-sized blocks
                per = t.numel() // B
                return t.view(-1)[lo * per:hi * per]
            ts = [[rows(t) for t in grp] for grp in (lbs, ubs, duals, prim)]
            tabs = [(C.c_void_p * len(g))(*[t.data_ptr() for t in g]) for g in ts]
            batch = _lib.Batch(tabs[0], tabs[1], tabs[2], tabs[3], rows(x_lp).data_ptr(), rows(pw).data_ptr(),
                               rows(pb).data_ptr(), rows(mask).data_ptr(), len(lbs), len(duals), len(prim))
            ws = self.workspace(n, c)
            rc = self.lib.gnnb_forward(self.h, C.byref(batch), n, scores[lo:hi].data_ptr(), dec[lo:hi].data_ptr(),
                                       status[c:c + 1].data_ptr(), ws.data_ptr(), ws.numel(), C.c_void_p(stream_ptr))
            _lib.check(rc, "gnnb_forward")

        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            if nchunk == 1:
                launch(0, cur.cuda_stream)
            else:
                while len(self._streams) < nchunk:
                    self._streams.append(torch.cuda.Stream(device=self.device))
                for c in range(nchunk):
                    st = self._streams[c]
                    st.wait_stream(cur)                   # inputs were produced on the caller's stream
                    launch(c, st.cuda_stream)
                for c in range(nchunk):
                    # results are ordered back into the caller's stream; since every tensor used here belongs to that
                    # stream and it now waits for the side streams, the caching allocator cannot recycle them early
                    cur.wait_stream(self._streams[c])
        return ForwardResult(scores, dec, status, mask2)

    # ---- the reference's own call pattern: host tensors, one or two subproblems ---------------------------------
    @staticmethod
    def _host(t):
        """float32 C-contiguous numpy view / copy of a tensor (of ANY device: the reference's driver hands over `.cuda()`
        layers next to CPU bounds, relu_conv_gnnkwthreshold.py:111-117), python list or array -- always HOST memory."""
        if torch.is_tensor(t):
            t = t.detach()
            if t.device.type != "cpu":
                t = t.cpu()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(torch.float32).contiguous()
            return t.numpy()
        if type(t) is list and t and type(t[0]) is float:
            # the LP primals arrive as flat python lists of floats (graph_score.py:30): array('f') walks them in C, a third faster
            # than numpy's generic sequence path (same round-to-nearest double -> float conversion)
            try:
                return np.frombuffer(_array("f", t), dtype=np.float32)
            except TypeError:
                pass
        return np.ascontiguousarray(t, dtype=np.float32)

    class _HostBuf:
        """address + element count of a float32 C-contiguous host buffer (a CPU tensor as it is, or a numpy copy of a python list /
        array / tensor of another layout); holds the owner alive.  (A decision is ~0.3 ms of device work: numpy views and
        ``.ctypes`` objects for two dozen small inputs were a tenth of that again.)"""
        __slots__ = ("ptr", "size", "keep")

        def __init__(self, t):
            # data_ptr() is taken as a HOST address only of a tensor that lives on the host: a device tensor slipping through
            # here would be a wild host read inside gnnb_forward_host, so it is copied back by _host instead
            if torch.is_tensor(t) and t.device.type == "cpu" and t.dtype == torch.float32 and t.is_contiguous() \
                    and not t.requires_grad:
                self.ptr, self.size, self.keep = t.data_ptr(), t.numel(), t
            else:
                a = ScorerEngine._host(t)
                self.ptr, self.size, self.keep = a.ctypes.data, a.size, a

    def _prop_host(self, props):
        """(B, N_L) weights and (B,) biases of the property layers as numpy arrays, cached on the layer objects' identity + version"""
        key = tuple((id(l), l.weight.data_ptr(), l.weight._version, l.bias._version) for l in props)
        hit = self._prop_host_cache
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        for l in props:
            if l.weight.shape[0] != 1:
                raise NotImplementedError("the property layer must be Linear(., 1)")   # graph_conv.py:80
        pw = np.ascontiguousarray(np.stack([self._host(l.weight)[0] for l in props]))
        pb = np.ascontiguousarray(np.array([self._host(l.bias)[0] for l in props], dtype=np.float32))
        self._prop_host_cache = (key, pw, pb, list(props))
        return pw, pb

    def forward_host(self, lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, layers, masks, want_scores=False):
        """``forward`` for CPU inputs through ``gnnb_forward_host``: every input the kernels read is packed into ONE pinned
        transfer by the library, and decisions / status (/ scores) come back in one block -- no torch device tensors, no
        per-tensor ``.cuda()`` (graph_score.py:26-30).  Synchronous.  Returns (decisions (B, 2) int32 array, scores (B, R)
        float32 array or None); raises FloatingPointError like ``ForwardResult.check``."""
        fixed = layers["fixed_layers"]
        self.bind(fixed, tuple(lower_bounds_all[0].shape[1:]))
        B = int(lower_bounds_all[0].shape[0])
        if len(layers["prop_layers"]) != B:
            raise ValueError(f"{len(layers['prop_layers'])} property layers for a batch of {B}")
        HB = self._HostBuf
        lbs = [HB(t) for t in lower_bounds_all]
        ubs = [HB(t) for t in upper_bounds_all]
        duals = [HB(t) for t in dual_vars]
        prim = [HB(t) for t in primals]
        x_lp, mask = HB(primal_inputs), HB(masks)
        ng = len(self.sizes)
        if len(lbs) != ng or len(ubs) != ng:
            raise ValueError(f"{len(lbs)} bound tensors, layer graph has {ng} layers")
        for k, (l, u) in enumerate(zip(lbs, ubs)):
            if l.size != B * self.sizes[k] or u.size != B * self.sizes[k]:
                raise ValueError(f"bounds of graph layer {k}: {l.size} values, expected {B}x{self.sizes[k]}")
        for k, d in enumerate(duals):
            if d.size != B * self.sizes[k + 1] * 3:
                raise ValueError(f"dual_vars[{k}] has {d.size} values, expected ({B * self.sizes[k + 1]}, 3)")
        if mask.size != B * self.R:
            raise ValueError(f"masks has {mask.size} values, expected ({B}, {self.R})")
        if x_lp.size != B * self.sizes[0]:
            raise ValueError("primal_inputs has the wrong size")
        self._check_primals(fixed, prim, B)
        pw, pb = self._prop_host(layers["prop_layers"])
        tabs = [(C.c_void_p * len(g))(*[a.ptr for a in g]) for g in (lbs, ubs, duals, prim)]
        batch = _lib.Batch(tabs[0], tabs[1], tabs[2], tabs[3], x_lp.ptr, pw.ctypes.data, pb.ctypes.data, mask.ptr,
                           len(lbs), len(duals), len(prim))
        dec = np.empty((B, 2), dtype=np.int32)
        status = np.zeros(1, dtype=np.int32)
        scores = np.empty((B, self.R), dtype=np.float32) if want_scores else None
        if torch.cuda.current_device() == self._dev_index:
            st = torch.cuda.current_stream().cuda_stream
            rc = self.lib.gnnb_forward_host(self.h, C.byref(batch), B, scores.ctypes.data if want_scores else None, dec.ctypes.data,
                                            status.ctypes.data, C.c_void_p(st))
        else:
            with torch.cuda.device(self.device):
                st = torch.cuda.current_stream().cuda_stream
                rc = self.lib.gnnb_forward_host(self.h, C.byref(batch), B, scores.ctypes.data if want_scores else None, dec.ctypes.data,
                                                status.ctypes.data, C.c_void_p(st))
        _lib.check(rc, "gnnb_forward_host")
        _raise_for_status(int(status[0]))
        return dec, scores

    # ---- online learning --------------------------------------------------------
    def get_weights(self):
        """The GNN parameters as a flat float32 array in checkpoint order (see state_blob)."""
        out = np.empty(GNN_BLOB_FLOATS, dtype=np.float32)
        _lib.check(self.lib.gnnb_get_weights(self.h, out.ctypes.data_as(C.c_void_p), out.size), "gnnb_get_weights")
        return out

    def set_weights(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gnnb_set_weights(self.h, blob.ctypes.data_as(C.c_void_p), blob.size), "gnnb_set_weights")
        self._blob = blob

    def online_create(self, lr=1e-4, wd=1e-4):
        """torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd) of graph_score_online.py:15."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gnnb_online_create(self.h, lr, wd), "gnnb_online_create")
        self._online = True

    def online_step(self, args, kw_index, improvement, apply=True, want_scores=False):
        """graph_score_online.py:62-77 for the batch ``args`` (the argument tuple of GraphNet.forward): kw_index (B) flat
        indices into the R ReLU nodes, improvement (B).  Returns (loss (B) numpy, scores (B, R) device tensor or None)."""
        if not getattr(self, "_online", False):
            raise RuntimeError("online_step: call online_create first")
        B, lbs, ubs, duals, prim, x_lp, mask, pw, pb = self._marshal(*args)
        kw = np.ascontiguousarray(kw_index, dtype=np.int32).reshape(-1)
        imp = np.ascontiguousarray(improvement, dtype=np.float32).reshape(-1)
        if kw.size != B or imp.size != B:
            raise ValueError(f"online_step: {kw.size} KW decisions / {imp.size} improvements for a batch of {B}")
        mask_host = mask.view(B, self.R).cpu()
        for b in range(B):
            if not (0 <= kw[b] < self.R) or mask_host[b, kw[b]] == 0:
                raise IndexError(f"online_step: KW decision {int(kw[b])} of subproblem {b} is not an undecided ReLU of its mask")
        loss = np.empty(B, dtype=np.float32)
        scores = torch.empty(B, self.R, dtype=torch.float32, device=self.device) if want_scores else None
        tabs = [(C.c_void_p * len(g))(*[t.data_ptr() for t in g]) for g in (lbs, ubs, duals, prim)]
        batch = _lib.Batch(tabs[0], tabs[1], tabs[2], tabs[3], x_lp.data_ptr(), pw.data_ptr(), pb.data_ptr(), mask.data_ptr(),
                           len(lbs), len(duals), len(prim))
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream().cuda_stream
            rc = self.lib.gnnb_online_step(self.h, C.byref(batch), B, kw.ctypes.data_as(C.c_void_p), imp.ctypes.data_as(C.c_void_p),
                                           loss.ctypes.data_as(C.c_void_p), scores.data_ptr() if want_scores else None,
                                           1 if apply else 0, C.c_void_p(st))
        _lib.check(rc, "gnnb_online_step")
        return loss, scores

    def online_grad(self):
        """d loss / d parameters of the last online_step, flat float32 array in checkpoint order."""
        out = np.empty(GNN_BLOB_FLOATS, dtype=np.float32)
        _lib.check(self.lib.gnnb_online_grad(self.h, out.ctypes.data_as(C.c_void_p), out.size), "gnnb_online_grad")
        return out

    # ---- BaBSR fallback scorer -------------------------------------------------
    def babsr(self, lower_bounds_all, upper_bounds_all, layers, masks):
        """kw_score_conv.py choose_node_conv :41-113 for a batch: same bounds / layers / masks arguments as
        ``forward``; returns a BabsrResult (device tensors, no synchronisation)."""
        fixed = layers["fixed_layers"]
        self.bind(fixed, tuple(lower_bounds_all[0].shape[1:]))
        B = int(lower_bounds_all[0].shape[0])
        if len(layers["prop_layers"]) != B:
            raise ValueError(f"{len(layers['prop_layers'])} property layers for a batch of {B}")
        ng = len(self.sizes)
        if len(lower_bounds_all) != ng or len(upper_bounds_all) != ng:
            raise ValueError(f"{len(lower_bounds_all)} bound tensors, layer graph has {ng} layers")
        lbs = [self._dev(t) for t in lower_bounds_all]
        ubs = [self._dev(t) for t in upper_bounds_all]
        for k, (l, u) in enumerate(zip(lbs, ubs)):
            if l.numel() != B * self.sizes[k] or u.numel() != B * self.sizes[k]:
                raise ValueError(f"bounds of graph layer {k}: {tuple(l.shape)} does not hold {B}x{self.sizes[k]} values")
        mask = self._dev(masks)
        if mask.numel() != B * self.R:
            raise ValueError(f"masks has {tuple(mask.shape)}, expected ({B}, {self.R})")
        pw, _ = self._prop(layers["prop_layers"])
        scores = torch.empty(B, self.R, dtype=torch.float32, device=self.device)
        icp = torch.empty(B, self.R, dtype=torch.float32, device=self.device)
        tl = (C.c_void_p * ng)(*[t.data_ptr() for t in lbs])
        tu = (C.c_void_p * ng)(*[t.data_ptr() for t in ubs])
        with torch.cuda.device(self.device):
            rc = self.lib.gnnb_babsr(self.h, tl, tu, ng, pw.data_ptr(), mask.data_ptr(), B, scores.data_ptr(),
                                     icp.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "gnnb_babsr")
        return BabsrResult(scores, icp, mask.view(B, self.R), self.sizes[1:-1])

    def _check_primals(self, fixed, prim, B):
        def count(t):
            return t.numel() if torch.is_tensor(t) else t.size      # (tensors, numpy arrays, _HostBuf)
        if len(prim) != len(fixed) + 1:
            raise ValueError(f"{len(prim)} primal tensors for {len(fixed) + 1} network layers")
        k = 0
        for q, l in enumerate(fixed):
            if type(l) is nn.ReLU:
                k += 1
                n = B * self.sizes[k]
                if count(prim[q - 1]) != n or count(prim[q]) != n:
                    raise ValueError(f"primals[{q - 1}], primals[{q}] must hold {n} values each")
        if count(prim[-1]) != B:
            raise ValueError("primals[-1] must hold one value per subproblem")

    # ---- inspection (tests / bench) ---------------------------------------------------------
    def mu_rows(self, B, k):
        """(rows, linear_id): the raw rows of graph layer k as the last forward at batch B left them in the workspace -- a
        (B, N_k, p) view -- and the index (state-dict order) of the Linear they still have to go through: producers leave
        their last Linear to the consumer, mu = (W.rows + b).[r0 != 0]; -1: the rows are final.
        Rows of dead nodes, and after a default (restricted) forward the rows of layer 1 that are not scored, hold
        whatever was there before."""
        off, n = C.c_size_t(), C.c_size_t()
        _lib.check(self.lib.gnnb_mu_location(self.h, B, k, C.byref(off), C.byref(n)), "gnnb_mu_location")
        if self.n_streams > 1 and B >= self.n_streams * self.min_chunk:
            raise RuntimeError("mu() inspects a single-chunk forward: set engine.n_streams = 1 first")
        ws = self.workspace(B)
        rows = ws[off.value:off.value + 4 * n.value].view(torch.float32).view(B, self.sizes[k], self.p)
        lid = C.c_int(-1)
        _lib.check(self.lib.gnnb_mu_projection(self.h, k, C.byref(lid)), "gnnb_mu_projection")
        return rows, lid.value

    def linear_host(self, idx):
        """(weight (out, in), bias (out)) of the idx-th Linear of the checkpoint as float64 numpy arrays."""
        off = sum(o * i + o for o, i in GNN_LINEARS[:idx])
        o, i = GNN_LINEARS[idx]
        return (self._blob[off:off + o * i].reshape(o, i).astype(np.float64), self._blob[off + o * i:off + o * i + o].astype(np.float64))

    def mu(self, B, k):
        """Embedding mu[k] (B, N_k, p) of the last forward at batch B, for inspection: the deferred projection is applied
        on the HOST in float64 (numpy), nothing of it runs through torch on the GPU.  Returns a float32 CPU tensor."""
        rows, lid = self.mu_rows(B, k)
        rows = rows.cpu().numpy()
        if lid < 0:
            return torch.from_numpy(rows.copy())
        W, b = self.linear_host(lid)
        live = np.ones(rows.shape[:2], dtype=bool)
        if 1 <= k < len(self.sizes) - 1:
            lb, ub = (t.reshape(B, -1).cpu().numpy() for t in self._last_bounds[k])
            lower_temp, upper_temp = lb - np.maximum(lb, 0), np.maximum(ub, 0)
            with np.errstate(divide="ignore", invalid="ignore"):
                live = (upper_temp / (upper_temp - lower_temp)) != 0      # graph_conv.py:178 / :347
        safe = np.where(live[..., None], rows, 0.0).astype(np.float64)     # dead rows may never have been written
        out = (safe @ W.T + b) * live[..., None]
        return torch.from_numpy(out.astype(np.float32))

    def describe(self):
        """The launch plan of one forward on the bound network (dict parsed from gnnb_describe's JSON)."""
        import json
        buf = C.create_string_buffer(1 << 16)
        _lib.check(self.lib.gnnb_describe(self.h, buf, len(buf)), "gnnb_describe")
        return json.loads(buf.value.decode())

    def set_halfpass_limit(self, n):
        _lib.check(self.lib.gnnb_set_halfpass_limit(self.h, n), "gnnb_set_halfpass_limit")

    def profile_enable(self, on):
        _lib.check(self.lib.gnnb_profile_enable(self.h, int(on)), "gnnb_profile_enable")

    def profile_trace(self, cap=4096):
        """[(kernel class name, ms)] of the launches ``profile_read`` has resolved since the last call, in launch order."""
        cls, ms = (C.c_int * cap)(), (C.c_double * cap)()
        n = self.lib.gnnb_profile_trace(self.h, cls, ms, cap)
        if n < 0:
            raise RuntimeError("gnnb_profile_trace failed")
        return [(self.lib.gnnb_profile_class_name(cls[i]).decode(), ms[i]) for i in range(min(n, cap))]

    def profile_read(self, reset=True):
        n = self.lib.gnnb_profile_classes()
        ms, cnt = (C.c_double * n)(), (C.c_int64 * n)()
        _lib.check(self.lib.gnnb_profile_read(self.h, ms, cnt, n, int(reset)), "gnnb_profile_read")
        return {self.lib.gnnb_profile_class_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}
