"""SparseModule / SparseSequential (spconv/modules.py upstream; used at spconv_backbone.py:20,29,76-116)."""
from collections import OrderedDict

import torch
from torch import nn

from . import ops
from .tensor import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base class: modules that consume and produce a SparseConvTensor."""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):
    """Sequential container mixing sparse modules with dense ones (BatchNorm1d, ReLU, ...) that act on `.features`."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError('index {} is out of range'.format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        if not torch.is_grad_enabled():
            return self._forward_inference(input)
        for module in self._modules.values():
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:        # BatchNorm1d cannot take an empty batch
                    input.features = module(input.features)
            else:
                input = module(input)
        return input


def _forward_inference(self, input):
    """no-grad path: a sparse convolution followed by an eval-mode BatchNorm1d (and a ReLU) runs as ONE launch with the
    normalisation folded into the weights and the activation in the GEMM epilogue (conv.forward_fused)."""
    from .conv import SparseConvolution
    mods = list(self._modules.values())
    i = 0
    while i < len(mods):
        module = mods[i]
        if isinstance(module, SparseConvolution) and isinstance(input, SparseConvTensor) and input.indices.shape[0] != 0:
            bn = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(bn, nn.BatchNorm1d) and not bn.training and bn.track_running_stats:
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                input = module.forward_fused(input, bn, relu)
                i += 3 if relu else 2
                continue
        if is_spconv_module(module):
            assert isinstance(input, SparseConvTensor)
            input = module(input)
        elif isinstance(input, SparseConvTensor):
            if input.indices.shape[0] != 0:
                input.features = module(input.features)
        else:
            input = module(input)
        i += 1
    return input


SparseSequential._forward_inference = _forward_inference


def _sparse_convs(module):
    """sparse convolutions of `module` in execution order (SparseSequential nests and composite SparseModules exposing
    their sparse children as attributes in execution order)"""
    from .conv import SparseConvolution
    if isinstance(module, (list, tuple)):
        for m in module:
            yield from _sparse_convs(m)
    elif isinstance(module, SparseConvolution):
        yield module
    elif isinstance(module, (SparseModule, nn.Sequential)):
        for child in module._modules.values():
            if isinstance(child, (SparseModule, nn.Sequential)):
                yield from _sparse_convs(child)


def prebuild_rulebooks(module, indices, spatial_shape, batch_size, indice_dict, pending=None):
    """Builds every rulebook of `module` (a module or a list of modules run back to back) from the coordinates alone and
    returns the (indices, spatial_shape) leaving it.
    Dense layers (BatchNorm1d, ReLU) are skipped.  Scheduling: a strided conv's rulebook needs one host read-back (the
    number of output sites); when a SubM layer is about to build a table for the level that strided conv consumes, the
    strided conv's output-site search is enqueued FIRST, so the SubM table build runs while the host waits for the count."""
    convs = list(_sparse_convs(module))
    pending = dict(pending or {})               # {id(conv): state of an output-site search begun by the caller}
    if indices.dtype != torch.int32:
        indices = indices.int()
    indices = indices.contiguous()
    for i, conv in enumerate(convs):
        if conv.subm and conv.needs_new_rulebook(indice_dict) and indices.shape[0] > 0:
            for nxt in convs[i + 1:]:
                if nxt.conv1x1 or nxt.subm:
                    continue                      # stays on this level
                if not nxt.inverse and nxt.needs_new_rulebook(indice_dict) and id(nxt) not in pending:
                    pending[id(nxt)] = nxt.begin_rulebook(indices, spatial_shape, batch_size, indice_dict)
                break
        indices, spatial_shape = conv.build_rulebook(indices, spatial_shape, batch_size, indice_dict, pending.pop(id(conv), None))
    return indices, spatial_shape


_RULEBOOK_STREAMS = {}
# stages the rulebook stream is enqueued ahead of the feature stream (0-3 measured equal: the host is not the limit any more)
RULEBOOK_STAGES_AHEAD = int(__import__("os").environ.get("LIDAR_SPCONV_AHEAD", "1"))


def _hand_tables_to(stream, indice_dict):
    """tables are allocated on the rulebook stream and read on the feature stream: tell the caching allocator"""
    for name, datas in indice_dict.items():
        if name.startswith("__") or datas.get("__handed__") is stream:
            continue
        datas["__handed__"] = stream
        for key in ("nbr", "nbr_t", "in_indices", "out_indices"):
            t = datas.get(key)
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)


def _trim_padding(outs, idict, true_rows):
    """capacity-sized tensors of a finished forward -> exact ones (prefix views: padding rows sit at the tail)"""
    def cut(t):
        n = true_rows.get(t.data_ptr()) if torch.is_tensor(t) else None
        return t if n is None or n >= t.shape[0] else t[:n]
    for x in outs:
        n = true_rows.get(x.indices.data_ptr())
        if n is not None and n < x.indices.shape[0]:
            x.features, x.indices = x.features[:n], x.indices[:n]
    for name, datas in idict.items():
        if name.startswith("__"):
            continue
        n_in, n_out = true_rows.get(datas["in_indices"].data_ptr()), true_rows.get(datas["out_indices"].data_ptr())
        if n_in is None and n_out is None:
            continue
        if n_out is not None:
            datas["nbr"] = datas["nbr"][:n_out]
        if datas.get("nbr_t") is not None and n_in is not None:
            datas["nbr_t"] = datas["nbr_t"][:n_in]
        if datas["subm"]:
            datas["nbr_t"] = datas["nbr"]
        datas["in_indices"], datas["out_indices"] = cut(datas["in_indices"]), cut(datas["out_indices"])
        datas.pop("order", None)                 # the row orders cover the padding rows too: recomputed if anyone asks again
        datas.pop("order_t", None)


def run_stages_pipelined(stages, x, speculate=True):
    """Inference over a list of stages (modules run back to back): -> the output of every stage.
    The rulebooks depend on coordinates alone, so stage s+1's tables are built on a second stream WHILE stage s's GEMMs run.
    No host read-back happens inside the forward: tables of the strided convolutions are sized from what the same
    convolution produced last time (ops._finish_speculative: capacity + padding rows), and the true output counts are checked
    ONCE after the last launch has been enqueued; the outputs are then trimmed to their exact row counts.  A count that
    exceeded its capacity (or a first call, which has no history) runs the exact path, one read-back per strided convolution."""
    if speculate and ops.SPECULATE:
        entry = (x.features, x.indices, dict(x.indice_dict))
        spec = x.indice_dict["__spec__"] = ops.new_speculation()
        x.indice_dict["__inference__"] = True
        outs = _run_stages_pipelined(stages, x)
        true_rows, over = ops.resolve_speculation(spec)
        x.indice_dict.pop("__spec__", None)
        if not over:
            _trim_padding(outs, x.indice_dict, true_rows)
            return outs
        ops.GRIDS.reset()                        # candidate ids beyond a capacity are still in a grid: start from clean ones
        x.features, x.indices = entry[0], entry[1]
        x.indice_dict.clear()
        x.indice_dict.update(entry[2])
    x.indice_dict["__inference__"] = True
    return _run_stages_pipelined(stages, x)


def _run_stages_pipelined(stages, x):
    """Inference over a list of stages (modules run back to back): -> the output of every stage.
    The rulebooks depend on coordinates alone, so stage s+1's tables are built on a second stream WHILE stage s's GEMMs run:
    the host starts the next stage's output-site search, enqueues this stage's feature launches, and only then blocks on the
    search's count read-back — the hash builds, table fills and mask orders hide under the matrix work instead of preceding
    it (prebuild_rulebooks alone serialises ~1.3 ms of table building in front of a SECOND backbone's 2.1 ms of GEMMs)."""
    dev = x.features.device
    feat = torch.cuda.current_stream(dev)
    rb = _RULEBOOK_STREAMS.get(dev)
    if rb is None:
        rb = _RULEBOOK_STREAMS[dev] = torch.cuda.Stream(dev, priority=-1)     # the rulebook chain is the critical path of the forward: its
                                                                            # short kernels must not queue behind the GEMM workgroups
    indices = x.indices if x.indices.dtype == torch.int32 else x.indices.int()
    indices = indices.contiguous()
    x.indices = indices
    rb.wait_stream(feat)                         # the coordinates are produced on the feature stream
    indices.record_stream(rb)
    bs, idict = x.batch_size, x.indice_dict
    outs = []
    ahead = RULEBOOK_STAGES_AHEAD
    try:                                         # plain set_stream calls: the context manager costs the host ~10 us a time
        torch.cuda.set_stream(rb)
        ready, built, nxt = [], 0, (indices, x.spatial_shape)

        def build_next():
            nonlocal built, nxt
            nxt = prebuild_rulebooks(stages[built], nxt[0], nxt[1], bs, idict)
            ev = torch.cuda.Event()
            ev.record(rb)
            ready.append(ev)
            built += 1
            _hand_tables_to(feat, idict)

        while built < min(len(stages), 1 + ahead):
            build_next()
        for s, stage in enumerate(stages):
            torch.cuda.set_stream(feat)
            feat.wait_event(ready[s])
            x = stage(x)
            outs.append(x)
            torch.cuda.set_stream(rb)
            if built < len(stages):
                build_next()
    except BaseException:
        # a forward abandoned half-way (OOM in a stage, KeyboardInterrupt, LidarHipError) can leave candidate ids in an output-level
        # grid that no wipe will ever visit: forget the grids, new ones are initialised on demand
        torch.cuda.set_stream(feat)
        ops.GRIDS.reset()
        raise
    finally:
        torch.cuda.set_stream(feat)
    return outs
