"""Rulebook construction and the sparse-conv autograd function (host side of csrc/sparse_conv.hip).

upstream equivalents: spconv.ops.get_indice_pairs / get_conv_output_size, spconv.functional.indice_conv /
indice_subm_conv / indice_inverse_conv."""
import torch
from torch.autograd import Function

from .. import _lib, workspace


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation=None):
    return [(i + 2 * p - k) // s + 1 for i, k, s, p in zip(input_size, kernel_size, stride, padding)]


def _hash_table(indices, spatial_shape):
    L = _lib.lib()
    n = indices.shape[0]
    cap = L.lidar_spconv_hash_capacity(n)
    table = torch.empty(cap * 12, dtype=torch.uint8, device=indices.device)
    D, H, W = spatial_shape
    _lib.check(L.lidar_spconv_build_hash(_lib.ptr(indices), n, D, H, W, _lib.ptr(table), cap, _lib.stream()), "lidar_spconv_build_hash")
    return table, cap


# ---------------------------------------------------------------------------------------------- dense index grids
class GridPool:
    """Persistent (batch, D, H, W) int32 index grids, one per resolution level (csrc/rulebook_grid.hip): a coordinate lookup of
    the rulebook builders is then one load instead of a hash probe sequence.  A grid is allocated and set to "empty" once, holds
    the rows of ONE coordinate tensor at a time, and is wiped by revisiting exactly the cells that were written (the pool keeps
    its own copy of those coordinates: the caller's tensor may be a reused buffer).  Which tensor a grid holds is tracked per
    forward (`token`, kept in the rulebook dict), so a level's grid is scattered once and then serves the SubM table of the level,
    the strided convolution leaving it and the one entering it.  Levels too large for the budget fall back to the hash builder."""
    MAX_BYTES_PER_GRID = int(__import__("os").environ.get("LIDAR_SPCONV_GRID_MAX_GB", "8")) << 30
    MAX_TOTAL_BYTES = int(__import__("os").environ.get("LIDAR_SPCONV_GRID_TOTAL_GB", "64")) << 30     # all grids of the pool together
    ENABLED = __import__("os").environ.get("LIDAR_SPCONV_GRID", "1") != "0"

    def __init__(self):
        self.grids = {}        # (device, B, D, H, W) -> [grid, held coords copy or None, n held, token, stream event, B, last use]
        self.counter = 0

    def new_token(self):
        self.counter += 1
        return self.counter

    def holds_rows(self):
        return any(ent[2] > 0 for ent in self.grids.values())

    def wipe_all(self):
        """every grid back to all-empty, on the current stream (a captured forward must begin and end with clean grids)"""
        for key, ent in self.grids.items():
            if ent[2] > 0:
                self._join(ent)
                self._wipe(ent, key[2:])
                self._mark(ent)

    def reset(self):
        """forget every grid (a forward was abandoned half-way: cells may hold values nobody will wipe); new ones are
        allocated and initialised on demand.  Rare: waits for the device, the old grids may still be in use on any stream."""
        torch.cuda.synchronize()
        self.grids.clear()

    def _evict_for(self, nbytes):
        """make room for a new grid of `nbytes` (another batch size or resolution: the last partial eval batch, a bs-1 demo after
        bs-16): least-recently-used grids go first, row-holding ones included — after every forward each level's grid still holds
        its rows (wipes are lazy), so "empty grids only" would evict nothing.  A dropped grid needs no wipe (its memory is released;
        a level that needs one again gets a freshly initialised grid and scatters its rows again).  -> False when even an empty pool
        cannot take it within MAX_TOTAL_BYTES: the level then uses the hash builder.  Rare: waits for the device."""
        total = sum(e[0].numel() * 4 for e in self.grids.values())
        if total + nbytes <= self.MAX_TOTAL_BYTES:
            return True
        if nbytes > self.MAX_TOTAL_BYTES:
            return False
        torch.cuda.synchronize()                 # a grid may still be in use on any stream
        for key in sorted(self.grids, key=lambda k: self.grids[k][6]):
            total -= self.grids.pop(key)[0].numel() * 4
            if total + nbytes <= self.MAX_TOTAL_BYTES:
                break
        return total + nbytes <= self.MAX_TOTAL_BYTES

    def _entry(self, device, batch, shape):
        key = (str(device), int(batch), *[int(v) for v in shape])
        ent = self.grids.get(key)
        self.counter += 1
        if ent is not None:
            ent[6] = self.counter                # last use (LRU)
        if ent is None:
            cells = int(batch) * int(shape[0]) * int(shape[1]) * int(shape[2])
            if not self.ENABLED or cells * 4 > self.MAX_BYTES_PER_GRID or cells <= 0:
                return None
            if not self._evict_for(cells * 4):
                return None
            grid = torch.empty(cells, dtype=torch.int32, device=device)
            _lib.check(_lib.lib().lidar_spconv_grid_init(_lib.ptr(grid), cells, _lib.stream()), "lidar_spconv_grid_init")
            ent = self.grids[key] = [grid, None, 0, None, None, int(batch), self.counter]
        return ent

    @staticmethod
    def _join(ent):
        """the grid may last have been touched on another stream (rulebook stream vs the caller's): order after everything
        queued there.  Same stream as last time (the steady state: one rulebook stream) -> nothing to do, no event."""
        if ent[4] is not None and ent[4][0] != _lib.stream().value:
            torch.cuda.current_stream(ent[0].device).wait_stream(ent[4][1])

    @staticmethod
    def _mark(ent):
        sid = _lib.stream().value
        if ent[4] is None or ent[4][0] != sid:
            ent[4] = (sid, torch.cuda.current_stream(ent[0].device))

    def _rows(self, ent, coords, n, shape, mode):
        D, H, W = shape
        _lib.check(_lib.lib().lidar_spconv_grid_rows(_lib.ptr(coords), n, None, ent[5], D, H, W, _lib.ptr(ent[0]), mode, _lib.stream()),
                   "lidar_spconv_grid_rows")

    def _wipe(self, ent, shape):
        if ent[1] is not None and ent[2] > 0:
            if ent[1].is_cuda:
                ent[1].record_stream(torch.cuda.current_stream(ent[1].device))     # allocated on another stream, released right after
            self._rows(ent, ent[1], ent[2], shape, 1)
        ent[1], ent[2], ent[3] = None, 0, None

    def loaded(self, indices, batch, shape, token):
        """-> the level's grid holding the rows of `indices` (scattered now unless this forward already did), or None"""
        ent = self._entry(indices.device, batch, shape)
        if ent is None:
            return None
        self._join(ent)
        if ent[3] != (token, indices.data_ptr(), indices.shape[0]):
            self._wipe(ent, shape)
            keep = indices.clone()                                      # the cells written, for the wipe: never the caller's buffer
            keep.record_stream(torch.cuda.current_stream(keep.device))  # (and whichever stream wipes it later: _wipe records that one)
            self._rows(ent, keep, keep.shape[0], shape, 0)
            ent[1], ent[2], ent[3] = keep, keep.shape[0], (token, indices.data_ptr(), indices.shape[0])
        self._mark(ent)
        return ent[0]

    def emptied(self, device, batch, shape):
        ent = self._entry(device, batch, shape)
        if ent is None:
            return None
        self._join(ent)
        self._wipe(ent, shape)
        self._mark(ent)
        return ent

    def adopt_outputs(self, ent, out_indices, shape, token):
        """after lidar_spconv_grid_outputs: the grid holds candidate ids at exactly the cells of out_indices -> write the rows"""
        self._join(ent)
        self._rows(ent, out_indices, out_indices.shape[0], shape, 2)
        ent[1], ent[2], ent[3] = out_indices, out_indices.shape[0], (token, out_indices.data_ptr(), out_indices.shape[0])
        self._mark(ent)
        return ent[0]


GRIDS = GridPool()


def _grid_token(indice_dict):
    """one token per forward: lives in the rulebook dict that travels with the tensors"""
    if indice_dict is None:
        return GRIDS.new_token()
    t = indice_dict.get("__grid_token__")
    if t is None:
        t = indice_dict["__grid_token__"] = {"token": GRIDS.new_token()}
    return t["token"]


def _geom_args(spatial_shape, ksize, stride, padding):
    return [int(v) for v in (*spatial_shape, *ksize, *stride, *padding)]


def subm_rulebook(indices, spatial_shape, ksize, batch_size=None, indice_dict=None):
    """-> nbr (N, K) int32.  outputs == inputs.  With batch_size given the table is read off the level's dense index grid
    (GridPool); otherwise (or when the grid would not fit the budget) a coordinate hash table is built for it."""
    _lib.require_cuda(indices)
    n = indices.shape[0]
    K = ksize[0] * ksize[1] * ksize[2]
    nbr = torch.empty((n, K), dtype=torch.int32, device=indices.device)
    if n == 0:
        return nbr
    D, H, W = spatial_shape
    grid = GRIDS.loaded(indices, batch_size, spatial_shape, _grid_token(indice_dict)) if batch_size else None
    if grid is not None:
        pad = [k // 2 for k in ksize]
        _lib.check(_lib.lib().lidar_spconv_grid_table(_lib.ptr(indices), n, int(batch_size), *_geom_args(spatial_shape, ksize, (1, 1, 1), pad),
                                                      _lib.ptr(grid), n, _lib.ptr(nbr), _lib.stream()), "lidar_spconv_grid_table")
        return nbr
    table, cap = _hash_table(indices, spatial_shape)
    _lib.check(_lib.lib().lidar_spconv_subm_table(_lib.ptr(indices), n, D, H, W, ksize[0], ksize[1], ksize[2], _lib.ptr(table), cap,
                                                  _lib.ptr(nbr), _lib.stream()), "lidar_spconv_subm_table")
    return nbr


_PINNED = {"buf": None, "next": 0}


def _pinned_slot():
    """one int32 of pinned host memory for an asynchronous count read-back (a ring of 4096 slots: a slot comes round again
    thousands of rulebooks later, long after its count was read)"""
    if _PINNED["buf"] is None:
        _PINNED["buf"] = torch.zeros((4096,), dtype=torch.int32).pin_memory()
    i = _PINNED["next"]
    _PINNED["next"] = (i + 1) % 4096
    return _PINNED["buf"][i:i + 1]


def conv_rulebook_begin(indices, batch_size, spatial_shape, ksize, stride, padding, indice_dict=None):
    """Phase 1 of the SparseConv3d rulebook: enqueue the search for the unique output sites and an asynchronous copy of
    their count to pinned host memory.  Returns the pending state for conv_rulebook_finish; work enqueued on the stream
    after this call (e.g. the SubM table of the same level) overlaps the host's wait for the count."""
    _lib.require_cuda(indices)
    L = _lib.lib()
    n = indices.shape[0]
    K = ksize[0] * ksize[1] * ksize[2]
    dev = indices.device
    st = {"n": n, "K": K, "dev": dev, "ksize": list(ksize), "stride": list(stride), "padding": list(padding), "indices": indices,
          "shape": list(spatial_shape), "out_shape": get_conv_output_size(spatial_shape, ksize, stride, padding), "grid": None,
          "hint_key": (str(dev), int(batch_size), *_geom_args(spatial_shape, ksize, stride, padding))}
    if n == 0:
        return st
    bound = n
    for k, s in zip(ksize, stride):
        bound *= -(-k // s)
    bound = min(bound, n * K)
    out_idx = torch.empty((bound, 4), dtype=torch.int32, device=dev)
    num = torch.empty((1,), dtype=torch.int32, device=dev)
    D, H, W = spatial_shape
    token = _grid_token(indice_dict)
    # a convolution whose output level has the input level's shape (k 3, s 1, p 1) would share ONE pool grid between its input
    # rows and its output candidates (the pool is keyed by level shape): emptied() would wipe the rows just scattered and the
    # table builder would read output ids as input rows.  Such a level takes the hash builder.
    same_level = [int(v) for v in st["out_shape"]] == [int(v) for v in spatial_shape]
    gin = None if same_level else GRIDS.loaded(indices, batch_size, spatial_shape, token)
    gout = GRIDS.emptied(dev, batch_size, st["out_shape"]) if gin is not None else None
    if gin is not None and gout is not None and n * K <= 0x3FFFFFFF:
        wsb = L.lidar_spconv_grid_outputs_workspace_bytes(n, K)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.check(L.lidar_spconv_grid_outputs(_lib.ptr(indices), n, int(batch_size), *_geom_args(spatial_shape, ksize, stride, padding), _lib.ptr(gout[0]),
                                               _lib.ptr(out_idx), _lib.ptr(num), _lib.ptr(ws), wsb, _lib.stream()),
                   "lidar_spconv_grid_outputs")
        GridPool._mark(gout)
        st.update(grid=(gin, gout, token, batch_size))
    else:
        wsb = L.lidar_spconv_conv_table_workspace_bytes(n, *ksize, *stride)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)          # lives until finish (several rulebooks may be in flight)
        _lib.check(L.lidar_spconv_conv_outputs(_lib.ptr(indices), n, batch_size, D, H, W, *ksize, *stride, *padding, _lib.ptr(out_idx), bound,
                                               _lib.ptr(num), _lib.ptr(ws), wsb, _lib.stream()), "lidar_spconv_conv_outputs")
    num_host = _pinned_slot()
    num_host.copy_(num, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    st.update(ws=ws, wsb=wsb, out_idx=out_idx, num=num, num_host=num_host, ev=ev, bound=bound)
    return st


# ---- capacity-sized rulebooks: no host read-back inside a forward ---------------------------------------------------------
# The only thing the host ever needs from the output-site search is the NUMBER of output sites, to size the tables.  An
# inference pipeline (modules.run_stages_pipelined) instead sizes them from what the same convolution produced last time
# (outputs per input row, + headroom), marks the unused tail rows as padding rows on the device (batch index -1: no
# neighbours, reach nothing, lidar_spconv_grid_pad_rows) and checks all counts ONCE at the end of the forward, when the GPU
# still has the GEMMs to chew on.  A count above its capacity discards the forward and replays it on the exact path.
_CAP_HINTS = {}              # geometry key -> output sites per input site seen last time
CAP_HEADROOM, CAP_SLACK = 1.12, 2048
SPECULATE = __import__("os").environ.get("LIDAR_SPCONV_SPECULATE", "1") != "0"


def new_speculation():
    """state of one forward: rows = {coords data_ptr: (estimated true rows, index of the pending count or None)}"""
    return {"rows": {}, "pending": []}


def _finish_speculative(st, spec, need_t):
    hint = _CAP_HINTS.get(st["hint_key"])
    if hint is None or st["grid"] is None:
        return None
    n, K, dev = st["n"], st["K"], st["dev"]
    est_in, src = spec["rows"].get(st["indices"].data_ptr(), (float(n), None))
    est_out = est_in * hint
    cap = min(st["bound"], int(est_out * CAP_HEADROOM) + CAP_SLACK)
    if "caps" in spec:                                                  # frozen capacities (a captured graph): by position
        cap = min(st["bound"], spec["caps"][len(spec["pending"])])
    L = _lib.lib()
    out_indices = st["out_idx"][:cap]                                   # a view of this rulebook's own buffer
    _lib.check(L.lidar_spconv_grid_pad_rows(_lib.ptr(out_indices), _lib.ptr(st["num"]), cap, _lib.stream()), "lidar_spconv_grid_pad_rows")
    gin, gout, token, bsz = st["grid"]
    geom = [int(bsz)] + _geom_args(st["shape"], st["ksize"], st["stride"], st["padding"])
    go = GRIDS.adopt_outputs(gout, out_indices, st["out_shape"], token)
    nbr = torch.empty((cap, K), dtype=torch.int32, device=dev)
    _lib.check(L.lidar_spconv_grid_table(_lib.ptr(out_indices), cap, *geom, _lib.ptr(gin), n, _lib.ptr(nbr), _lib.stream()),
               "lidar_spconv_grid_table")
    nbr_t = None
    if need_t:
        nbr_t = torch.empty((n, K), dtype=torch.int32, device=dev)
        _lib.check(L.lidar_spconv_grid_table_t(_lib.ptr(st["indices"]), n, *geom, _lib.ptr(go), cap, _lib.ptr(nbr_t), _lib.stream()),
                   "lidar_spconv_grid_table_t")
    spec["pending"].append({"key": st["hint_key"], "num_host": st["num_host"], "ev": st["ev"], "cap": cap, "src": src,
                            "n_in": n if src is None and st["indices"].data_ptr() not in spec["rows"] else int(est_in),
                            "bound": st["bound"]})
    spec["rows"][out_indices.data_ptr()] = (est_out, len(spec["pending"]) - 1)
    return out_indices, nbr, nbr_t


def resolve_speculation(spec, wait=True):
    """Wait for the counts of this forward (one event: they are produced in order on one stream; wait=False: the caller has
    already synchronised), refresh the hints.
    -> (true row count per coords data_ptr, overflowed?)"""
    pend = spec["pending"]
    if not pend:
        return {}, False
    if wait:
        pend[-1]["ev"].synchronize()
    over = False
    for p in pend:
        p["true"] = int(p["num_host"][0])
        true_in = p["n_in"] if p["src"] is None else pend[p["src"]]["true"]
        if true_in > 0:
            _CAP_HINTS[p["key"]] = p["true"] / true_in
        over = over or p["true"] > p["cap"]
    return {ptr: pend[i]["true"] for ptr, (_, i) in spec["rows"].items() if i is not None}, over


def conv_rulebook_finish(st, spec=None, need_t=True):
    """Phase 2: allocate and fill the tables.  Exact path: wait for the count only (not for later work on the stream).
    spec (new_speculation()): size by the capacity hint instead, no wait (see above); need_t=False: no transposed table
    (inference never reads it; ensure_table_t builds it on demand)."""
    n, K, dev = st["n"], st["K"], st["dev"]
    if n == 0:
        return (torch.empty((0, 4), dtype=torch.int32, device=dev), torch.empty((0, K), dtype=torch.int32, device=dev),
                torch.empty((0, K), dtype=torch.int32, device=dev))
    if spec is not None and SPECULATE:
        done = _finish_speculative(st, spec, need_t)
        if done is not None:
            return done
    st["ev"].synchronize()
    n_out = int(st["num_host"][0])
    src = spec["rows"].get(st["indices"].data_ptr()) if spec is not None else None
    true_in = n if src is None else (int(src[0]) if src[1] is None else int(spec["pending"][src[1]]["num_host"][0]))   # (an earlier
                                                                                                                      # count of this stream)
    if true_in > 0:
        _CAP_HINTS[st["hint_key"]] = n_out / true_in
    nbr = torch.empty((n_out, K), dtype=torch.int32, device=dev)
    nbr_t = torch.empty((n, K), dtype=torch.int32, device=dev) if need_t else None
    out_indices = st["out_idx"][:n_out].clone()
    L = _lib.lib()
    if st["grid"] is not None:
        gin, gout, token, bsz = st["grid"]
        geom = [int(bsz)] + _geom_args(st["shape"], st["ksize"], st["stride"], st["padding"])
        go = GRIDS.adopt_outputs(gout, out_indices, st["out_shape"], token)       # the output level's grid: rows instead of candidates
        _lib.check(L.lidar_spconv_grid_table(_lib.ptr(out_indices), n_out, *geom, _lib.ptr(gin), n, _lib.ptr(nbr), _lib.stream()),
                   "lidar_spconv_grid_table")
        if need_t:
            _lib.check(L.lidar_spconv_grid_table_t(_lib.ptr(st["indices"]), n, *geom, _lib.ptr(go), n_out, _lib.ptr(nbr_t), _lib.stream()),
                       "lidar_spconv_grid_table_t")
        return out_indices, nbr, nbr_t
    if nbr_t is None:
        nbr_t = torch.empty((n, K), dtype=torch.int32, device=dev)                # the hash builder fills both in one pass
    _lib.check(L.lidar_spconv_conv_tables(n, *st["ksize"], *st["stride"], n_out, _lib.ptr(nbr), _lib.ptr(nbr_t), _lib.ptr(st["ws"]),
                                          st["wsb"], _lib.stream()), "lidar_spconv_conv_tables")
    return out_indices, nbr, nbr_t


def ensure_table_t(datas):
    """the transposed table of a regular convolution's rulebook, built from the forward table when the builder skipped it"""
    if datas.get("nbr_t") is None:
        nbr = datas["nbr"]
        n_in, K = datas["in_indices"].shape[0], nbr.shape[1]
        nbr_t = torch.full((n_in, K), -1, dtype=torch.int32, device=nbr.device)
        _lib.check(_lib.lib().lidar_spconv_transpose_table(_lib.ptr(nbr), nbr.shape[0], K, n_in, _lib.ptr(nbr_t), _lib.stream()),
                   "lidar_spconv_transpose_table")
        datas["nbr_t"] = nbr_t
    return datas["nbr_t"]


def conv_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, indice_dict=None):
    """-> out_indices (N_out, 4) int32, nbr (N_out, K), nbr_t (N_in, K).  One host read-back (N_out)."""
    try:
        return conv_rulebook_finish(conv_rulebook_begin(indices, batch_size, spatial_shape, ksize, stride, padding, indice_dict))
    except BaseException:
        GRIDS.reset()          # between begin and finish the output level's grid holds candidate ids no wipe would visit
        raise


def cached_mask_order(datas, key, table):
    """mask order of `table`, kept in the rulebook dict under `key` ([masks, order, event-or-None]); joins the side stream
    once when the order was scheduled there (conv._schedule_mask_order)."""
    st = datas.get(key)
    if st is None:
        st = datas[key] = list(mask_order(table)) + [None]
    if st[2] is not None:
        cur = torch.cuda.current_stream(table.device)
        cur.wait_event(st[2])
        st[0].record_stream(cur)                # computed on the side stream, read on this one
        st[1].record_stream(cur)
        st[2] = None
    return st[0], st[1]


def _implicit_gemm(feats, nbr, weight_kcc, bias, n_out, order=None):
    """out (n_out, Cout) = sum_k feats[nbr[:, k]] @ weight_kcc[k] (+ bias); `order` = mask_order(nbr) (same bits, faster)."""
    K, Cin, Cout = weight_kcc.shape
    out = torch.empty((n_out, Cout), dtype=torch.float32, device=feats.device)
    if n_out == 0:
        return out
    if order is not None:
        _lib.check(_lib.lib().lidar_spconv_implicit_gemm_sorted(_lib.ptr(feats), _lib.ptr(nbr), _lib.ptr(order[0]), _lib.ptr(order[1]),
                                                                n_out, K, Cin, Cout, _lib.ptr(weight_kcc), _lib.ptr(bias), None, 0,
                                                                _lib.ptr(out), _lib.stream()), "lidar_spconv_implicit_gemm_sorted")
        return out
    _lib.check(_lib.lib().lidar_spconv_implicit_gemm(_lib.ptr(feats), _lib.ptr(nbr), n_out, K, Cin, Cout, _lib.ptr(weight_kcc),
                                                     _lib.ptr(bias), _lib.ptr(out), _lib.stream()), "lidar_spconv_implicit_gemm")
    return out


def sorted_gemm_supported(K, Cin, Cout):
    return K <= 31 and bool(_lib.lib().lidar_spconv_sorted_gemm_supported(K, Cin, Cout))


_MG_WS = {}
_MG_RETIRED = []             # (workspace, event on its stream): regrown workspaces stay alive until their stream has passed the event


def _mask_group_workspace(n, device, stream):
    """persistent workspace of lidar_spconv_mask_group for ONE stream (its slot table must be empty between calls and two
    streams must not share one): grown with headroom, initialised once per buffer"""
    L = _lib.lib()
    need = int(L.lidar_spconv_mask_group_workspace_bytes(n))
    key = (device.index, int(stream.value or 0))
    ws = _MG_WS.get(key)
    if ws is None or ws.numel() < need:
        old = _MG_WS.get(key)
        if old is not None:                       # kernels on the side stream may still be using it: keep it alive until they are done
            _MG_RETIRED.append((old, torch.cuda.Event()))
            _MG_RETIRED[-1][1].record(torch.cuda.ExternalStream(int(stream.value or 0), device=device))
            while _MG_RETIRED and _MG_RETIRED[0][1].query():
                _MG_RETIRED.pop(0)
        ws = torch.empty(need + need // 4, dtype=torch.uint8, device=device)
        _lib.check(L.lidar_spconv_mask_group_init(_lib.ptr(ws), ws.numel(), stream), "lidar_spconv_mask_group_init")
        _MG_WS[key] = ws
    return ws


def mask_order(nbr, stream=None):
    """(n_out, K <= 31) neighbour table -> (row offset bit masks, a row order that groups equal masks, similar masks adjacent).
    Visiting rows in that order puts equal masks into the same MFMA tiles, so tiles stop multiplying padding rows and whole
    offsets drop out per workgroup (include/lidar_hip.h: lidar_spconv_mask_group, lidar_spconv_implicit_gemm_sorted)."""
    n_out, K = nbr.shape
    masks = torch.empty(n_out, dtype=torch.int32, device=nbr.device)
    order = torch.empty(n_out, dtype=torch.int32, device=nbr.device)
    if n_out:
        stream = _lib.stream() if stream is None else stream             # raw stream handle: the launches go there, whatever
        ws = _mask_group_workspace(n_out, nbr.device, stream)            # torch's current stream is (the caller orders the rest)
        _lib.check(_lib.lib().lidar_spconv_mask_group(_lib.ptr(nbr), n_out, K, _lib.ptr(masks), _lib.ptr(order), _lib.ptr(ws), ws.numel(),
                                                      stream), "lidar_spconv_mask_group")
    return masks, order


# A/B switch for the packed-weight GEMM kernel (LDS-DMA staging + 128-bit operand reads).  Default OFF: measured 8 % SLOWER than the
# register-A kernel on the SECOND stack (2.18 vs 2.01 ms, profiles/r04/spconv_gemm_layers.log) — W staging is not what holds that
# kernel back; kept for the bit-identity test and as the record of the experiment.
PACKED_GEMM = [__import__("os").environ.get("LIDAR_SPCONV_PACKED", "0") != "0"]


def pack_gemm_weights(weight_kcc):
    """(K, Cin, Cout) folded weights -> the lane-ordered packed form lidar_spconv_implicit_gemm_sorted_packed reads, or None when
    the shape has no packed kernel (csrc/sparse_conv.hip sc_implicit_gemm_pk_kernel)"""
    K, Cin, Cout = weight_kcc.shape
    L = _lib.lib()
    n = L.lidar_spconv_packed_floats(K, Cin, Cout)
    if n == 0 or not weight_kcc.is_cuda or not PACKED_GEMM[0]:
        return None
    w = weight_kcc.contiguous()
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.check(L.lidar_spconv_pack_weights(_lib.ptr(w), K, Cin, Cout, _lib.ptr(packed), _lib.stream()), "lidar_spconv_pack_weights")
    return packed


def indice_conv_fused(feats, nbr, weight_kcc, bias, residual=None, relu=False, order=None, packed=None):
    """Inference-only: act(sum_k feats[nbr[:, k]] @ weight_kcc[k] + bias + residual) in one launch (no autograd).
    order = mask_order(nbr) visits the rows in mask order (bit-identical result, far fewer padding MFMA tiles).
    packed = pack_gemm_weights(weight_kcc): the mask-ordered GEMM then runs its packed-weight kernel (bit-identical again)."""
    K, Cin, Cout = weight_kcc.shape
    n_out = nbr.shape[0]
    _lib.require_cuda(feats, weight_kcc, nbr, bias, residual)
    out = torch.empty((n_out, Cout), dtype=torch.float32, device=feats.device)
    if n_out == 0:
        return out
    if residual is not None and tuple(residual.shape) != (n_out, Cout):
        raise _lib.LidarHipError("indice_conv_fused: residual must be (n_out, Cout)")
    if order is not None and packed is not None and PACKED_GEMM[0]:
        masks, perm = order
        _lib.check(_lib.lib().lidar_spconv_implicit_gemm_sorted_packed(_lib.ptr(feats), _lib.ptr(nbr), _lib.ptr(masks), _lib.ptr(perm),
                                                                       n_out, K, Cin, Cout, _lib.ptr(packed), _lib.ptr(bias),
                                                                       _lib.ptr(residual), int(bool(relu)), _lib.ptr(out), _lib.stream()),
                   "lidar_spconv_implicit_gemm_sorted_packed")
        return out
    if order is not None:
        masks, perm = order
        _lib.check(_lib.lib().lidar_spconv_implicit_gemm_sorted(_lib.ptr(feats), _lib.ptr(nbr), _lib.ptr(masks), _lib.ptr(perm),
                                                                n_out, K, Cin, Cout, _lib.ptr(weight_kcc), _lib.ptr(bias),
                                                                _lib.ptr(residual), int(bool(relu)), _lib.ptr(out), _lib.stream()),
                   "lidar_spconv_implicit_gemm_sorted")
        return out
    _lib.check(_lib.lib().lidar_spconv_implicit_gemm_fused(_lib.ptr(feats), _lib.ptr(nbr), n_out, K, Cin, Cout,
                                                           _lib.ptr(weight_kcc), _lib.ptr(bias), _lib.ptr(residual), int(bool(relu)),
                                                           _lib.ptr(out), _lib.stream()), "lidar_spconv_implicit_gemm_fused")
    return out


class SparseConvFunction(Function):
    """features (N_in, Cin), weight (kD,kH,kW,Cin,Cout) [, bias] -> (N_out, Cout) through a neighbour table.

    fwd_table (N_out, K) drives the forward and the weight gradient; bwd_table (N_in, K) the input gradient.
    flip_bwd: submanifold tables are symmetric (bwd_table is fwd_table itself, read with the kernel flipped)."""

    @staticmethod
    def forward(ctx, features, weight, bias, fwd_table, bwd_table, flip_bwd, orders=None):
        """orders = (rulebook dict, key of the forward table's mask order, key of the backward table's) or None: the GEMMs
        then visit rows in mask order (bit-identical results, see lidar_spconv_implicit_gemm_sorted)."""
        feats = features.contiguous()
        Cin, Cout = weight.shape[-2], weight.shape[-1]
        w = weight.reshape(-1, Cin, Cout).contiguous()
        _lib.require_cuda(feats, w, fwd_table)
        order = None
        if orders is not None and fwd_table.shape[0] > 0 and sorted_gemm_supported(w.shape[0], Cin, Cout):
            order = cached_mask_order(orders[0], orders[1], fwd_table)
        out = _implicit_gemm(feats, fwd_table, w, bias.contiguous() if bias is not None else None, fwd_table.shape[0], order)
        ctx.save_for_backward(feats, w, fwd_table, bwd_table)
        ctx.flip_bwd, ctx.has_bias, ctx.wshape, ctx.orders = flip_bwd, bias is not None, weight.shape, orders
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feats, w, fwd_table, bwd_table = ctx.saved_tensors
        g = grad_out.contiguous()
        K, Cin, Cout = w.shape
        grad_feats = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            wt = w.flip(0) if ctx.flip_bwd else w
            order = None
            if ctx.orders is not None and bwd_table.shape[0] > 0 and sorted_gemm_supported(K, Cout, Cin):
                order = cached_mask_order(ctx.orders[0], ctx.orders[2], bwd_table)
            grad_feats = _implicit_gemm(g, bwd_table, wt.transpose(1, 2).contiguous(), None, feats.shape[0], order)
        if ctx.needs_input_grad[1]:
            L, n_out = _lib.lib(), fwd_table.shape[0]
            if n_out > 0 and L.lidar_spconv_wgrad_mfma_supported(K, Cin, Cout):
                order = None
                if ctx.orders is not None and K <= 31:
                    order = cached_mask_order(ctx.orders[0], ctx.orders[1], fwd_table)[1]
                grad_w = torch.empty_like(w)
                wsb = L.lidar_spconv_wgrad_workspace_bytes(n_out, K, Cin, Cout)
                ws = workspace.get("spconv_wgrad", wsb, g.device)
                _lib.check(L.lidar_spconv_wgrad_mfma(_lib.ptr(feats), _lib.ptr(g), _lib.ptr(fwd_table), _lib.ptr(order), n_out, K, Cin,
                                                     Cout, _lib.ptr(grad_w), _lib.ptr(ws), wsb, _lib.stream()), "lidar_spconv_wgrad_mfma")
            else:
                grad_w = torch.zeros_like(w)
                if n_out > 0:
                    _lib.check(L.lidar_spconv_wgrad(_lib.ptr(feats), _lib.ptr(g), _lib.ptr(fwd_table), n_out, K, Cin, Cout,
                                                    _lib.ptr(grad_w), _lib.stream()), "lidar_spconv_wgrad")
            grad_w = grad_w.view(ctx.wshape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_b = g.sum(0)
        return grad_feats, grad_w, grad_b, None, None, None, None


indice_conv = SparseConvFunction.apply
