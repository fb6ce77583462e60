"""A whole sparse inference forward as ONE hipGraph.

modules.run_stages_pipelined already runs a backbone without host read-backs (capacity-sized rulebooks, ops.new_speculation);
what is left on the host is ~80 kernel launches, events and allocations per forward — more host time than the first two
resolution levels take on the GPU, so those levels wait for the host.  With every size fixed (input rows padded to a capacity,
one frozen capacity per strided convolution) the same launches are captured once — rulebook stream, mask-order stream and
feature stream with their dependencies — and replayed with a single launch.  The true row counts arrive in pinned host
memory; a count above its capacity sends that forward through the exact path and the graph is rebuilt with more room.

Results are the exact ones (row counts, order, bits): padding rows are cut off before anything is returned.  The returned
tensors live in the graph's memory pool and are overwritten by the next call."""
import torch

from . import modules, ops
from .tensor import SparseConvTensor


class GraphedStages:
    def __init__(self, stages, spatial_shape, batch_size, channels, capacity, device, headroom=1.2):
        self.stages, self.shape, self.batch_size = list(stages), [int(v) for v in spatial_shape], int(batch_size)
        self.capacity, self.headroom = int(capacity), float(headroom)
        self.feats = torch.zeros((self.capacity, channels), dtype=torch.float32, device=device)
        self.coords = torch.full((self.capacity, 4), -1, dtype=torch.int32, device=device)
        self.stream = torch.cuda.Stream(device)
        self.graph = self.spec = self.outs = self.idict = self.key = None
        self.n = 0
        self.replays = self.fallbacks = 0

    # ------------------------------------------------------------------------------------------------ pieces
    def _load(self, feats, coords):
        n = feats.shape[0]
        if n > self.capacity or feats.shape[1] != self.feats.shape[1]:
            raise ValueError(f"GraphedStages: {tuple(feats.shape)} does not fit the static input {tuple(self.feats.shape)}")
        self.feats[:n].copy_(feats)
        self.coords[:n].copy_(coords)
        if self.n > n:
            self.coords[n:self.n].fill_(-1)              # rows of the previous, larger input -> padding rows
        self.n = n

    def _tensor(self):
        x = SparseConvTensor(self.feats, self.coords, self.shape, self.batch_size)
        x.indice_dict["__inference__"] = True
        return x

    def _eager(self, caps=None):
        """one capacity-sized forward on the static input, launched the ordinary way -> (outs, dict, spec)"""
        x = self._tensor()
        spec = x.indice_dict["__spec__"] = ops.new_speculation()
        spec["rows"][self.coords.data_ptr()] = (float(self.n), None)         # the input's true row count, for the hints
        if caps is not None:
            spec["caps"] = caps
        outs = modules._run_stages_pipelined(self.stages, x)
        return outs, x.indice_dict, spec

    def _exact(self):
        x = SparseConvTensor(self.feats[:self.n], self.coords[:self.n], self.shape, self.batch_size)
        return modules.run_stages_pipelined(self.stages, x, speculate=False)

    def _params_key(self):
        """the captured launches hold the addresses of the folded weights: any parameter or statistic that changed since -> new graph"""
        return tuple((t.data_ptr(), t._version) for st in self.stages for t in list(st.parameters()) + list(st.buffers()))

    def _capture(self):
        dev = self.feats.device
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._exact()                                                     # hints for every strided convolution (and warm-up)
            outs, idict, spec = self._eager()
            true_rows, over = ops.resolve_speculation(spec)
            if over:                                                          # cannot happen right after the exact pass
                raise RuntimeError("GraphedStages: capacity hints are inconsistent")
            caps = [min(p["bound"], int(p["true"] * self.headroom) + 4096) for p in spec["pending"]]
            self._eager(caps)                                                 # same sizes as the capture: every workspace exists
            ops.GRIDS.wipe_all()
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=self.stream):
                outs, idict, spec = self._eager(caps)
                here = torch.cuda.current_stream(dev)
                for pool in (modules._RULEBOOK_STREAMS, _side_streams()):
                    st = pool.get(dev)
                    if st is not None:
                        here.wait_stream(st)                                  # every branch joins before the capture ends
                ops.GRIDS.wipe_all()                                          # the next replay starts from clean grids again
        cur.wait_stream(self.stream)
        self.graph, self.spec, self.outs, self.idict, self.caps, self.key = graph, spec, outs, idict, caps, self._params_key()

    # ------------------------------------------------------------------------------------------------ call
    @torch.no_grad()
    def __call__(self, feats, coords):
        """feats (n, C), coords (n, 4) int32 [b, z, y, x], n <= capacity -> the output of every stage (exact row counts)"""
        self._load(feats, coords if coords.dtype == torch.int32 else coords.int())
        if self.graph is None or self.key != self._params_key():
            self._capture()
        if ops.GRIDS.holds_rows():
            ops.GRIDS.wipe_all()                                              # an eager forward left its rows behind
        self.graph.replay()
        torch.cuda.current_stream(self.feats.device).synchronize()            # the counts (pinned memory) are part of the graph
        self.replays += 1
        true_rows, over = ops.resolve_speculation(self.spec, wait=False)
        if over:                                                              # more output sites than a frozen capacity
            self.fallbacks += 1
            self.graph = None
            self.headroom *= 1.25
            ops.GRIDS.reset()
            return self._exact()
        true_rows[self.coords.data_ptr()] = self.n
        outs = [_view(x) for x in self.outs]
        idict = {k: (dict(v) if isinstance(v, dict) else v) for k, v in self.idict.items()}
        for x in outs:
            x.indice_dict = idict
        modules._trim_padding(outs, idict, true_rows)
        return outs


def _view(x):
    y = SparseConvTensor(x.features, x.indices, x.spatial_shape, x.batch_size)
    y.grid = x.grid
    return y


def _side_streams():
    from . import conv
    return conv._SIDE_STREAMS
