"""Student forward / backward on the MI355X (BASELINE cfg 4): ``encode_with_gradients``.

The reference's KD step (src/kd/train.py:176-210) calls ``StudentModel.encode_with_gradients``
twice per query (query, then its positive + hard negatives), multiplies, applies
``CombinedKDLoss`` and lets torch autograd walk back through the sentence-transformers modules.
Here the encoder forward that saves its activations and the whole backward pass are HIP
(``sskd_generic_forward`` / ``sskd_generic_backward``, csrc/train.hip) behind ONE
``torch.autograd.Function``; torch holds the fp32 master parameters (what the optimizer updates)
and supplies device memory.  bf16 compute, fp32 accumulation and gradients.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import _native
from .weights import BertConfig

_LAYER_MATS = (
    ("attention.self.query", "q"), ("attention.self.key", "k"), ("attention.self.value", "v"),
    ("attention.output.dense", "o"), ("intermediate.dense", "w1"), ("output.dense", "w2"),
)


def _pname(hf_name: str) -> str:
    return hf_name.replace(".", "__")


class _EncoderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: "TrainableEncoder", ids: torch.Tensor, mask: torch.Tensor, normalize: bool, *params):
        lib = _native.load()
        B, S = ids.shape
        module._check_autograd_contract()
        module._refresh_device_weights()
        need = int(lib.sskd_generic_workspace_bytes(module.cfg_struct, B, S, 1))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=ids.device)
        out = torch.empty((B, module.config.hidden_size), dtype=torch.float32, device=ids.device)
        with torch.cuda.device(ids.device):
            _native.check(lib.sskd_generic_forward(
                module.cfg_struct, module.w_struct, ids.data_ptr(), mask.data_ptr(), B, S, 1, 1, int(normalize),
                out.data_ptr(), ws.data_ptr(), ws.numel(), int(torch.cuda.current_stream(ids.device).cuda_stream)))
        ctx.module, ctx.ws, ctx.normalize = module, ws, bool(normalize)
        ctx.save_for_backward(ids, mask)
        ctx.weights_version = module._weights_version
        return out

    @staticmethod
    def backward(ctx, dout):
        """The HIP backward ACCUMULATES (atomics / +=) straight into ``p.grad`` of every parameter - views of ONE flat fp32
        buffer that is zeroed with one memset when the gradients were ``None`` - and hands autograd ``None`` for them:
        no per-call set of ~150 zero-filled buffers (90 MB for e5-small, the dense 30522 x 384 word-embedding gradient
        included) and no ~150 accumulation kernels when a step calls the encoder more than once (the reference calls
        ``encode_with_gradients`` 2 x batch_size times per step, src/kd/train.py:176-210).

        RESTRICTION (checked in ``TrainableEncoder._check_autograd_contract``, which raises instead of returning wrong
        gradients): this is ``loss.backward()`` semantics only.  Parameter gradients are side effects on ``p.grad``, so
        ``torch.autograd.grad(..., params)``, ``backward(inputs=...)``, tensor / post-accumulate hooks on a parameter
        (DDP registers those) and ``requires_grad=False`` on a single parameter are NOT served - the forward refuses
        parameters that carry hooks or are frozen.  Data parallelism reduces the FLAT gradient buffer instead
        (``TrainableEncoder.flat_grad``: one all-reduce of one tensor)."""
        lib = _native.load()
        module: TrainableEncoder = ctx.module
        ids, mask = ctx.saved_tensors
        B, S = ids.shape
        if ctx.weights_version != module._weights_version:
            raise RuntimeError("parameters changed between forward and backward of encode_with_gradients")
        module._attach_grads()
        dout = dout.to(torch.float32).contiguous()
        with torch.cuda.device(ids.device):
            _native.check(lib.sskd_generic_backward(
                module.cfg_struct, module.w_struct, module.g_struct, ids.data_ptr(), mask.data_ptr(), B, S, int(ctx.normalize),
                dout.data_ptr(), ctx.ws.data_ptr(), ctx.ws.numel(),
                int(torch.cuda.current_stream(ids.device).cuda_stream)))
        ctx.ws = None
        return (None, None, None, None) + (None,) * len(module.names)


class TrainableEncoder(nn.Module):
    """fp32 master parameters (HF ``BertModel`` names) + the bf16 device copies the kernels read.

    Every parameter is a VIEW of one flat fp32 buffer laid out the way the C-ABI wants it (per layer: Wq, Wk, Wv
    back to back = the fused [3H, H] QKV matrix, then their biases, ...), and every ``.grad`` a view of a second flat
    buffer with the same layout.  That makes the per-step housekeeping three launches instead of ~450: ONE cast of the
    flat master to bf16 + FOUR batched transposes (the backward's W^T operands, all 12 layers per launch) after an
    optimizer step, ONE memset of the gradients.  Optimizers see ordinary ``nn.Parameter``s."""

    def __init__(self, config: BertConfig, state_dict: Dict[str, np.ndarray], device, pos_offset: int = 0) -> None:
        super().__init__()
        _native.require_gpu()
        self.config = config
        self.device = torch.device(device)
        self.pos_offset = pos_offset
        H, F, L = config.hidden_size, config.intermediate_size, config.num_hidden_layers
        # ---- layout (element offsets into the flat buffers) ----
        layer_fields = [  # (hf suffix, shape)
            ("attention.self.query.weight", (H, H)), ("attention.self.key.weight", (H, H)), ("attention.self.value.weight", (H, H)),
            ("attention.self.query.bias", (H,)), ("attention.self.key.bias", (H,)), ("attention.self.value.bias", (H,)),
            ("attention.output.dense.weight", (H, H)), ("attention.output.dense.bias", (H,)),
            ("attention.output.LayerNorm.weight", (H,)), ("attention.output.LayerNorm.bias", (H,)),
            ("intermediate.dense.weight", (F, H)), ("intermediate.dense.bias", (F,)),
            ("output.dense.weight", (H, F)), ("output.dense.bias", (H,)),
            ("output.LayerNorm.weight", (H,)), ("output.LayerNorm.bias", (H,)),
        ]
        self._field_off: Dict[str, int] = {}
        off = 0
        for suffix, shape in layer_fields:
            self._field_off[suffix] = off
            off += int(np.prod(shape))
        self._layer_stride = off
        layout: Dict[str, tuple] = {}
        for i in range(L):
            for suffix, shape in layer_fields:
                layout[f"encoder.layer.{i}.{suffix}"] = (i * self._layer_stride + self._field_off[suffix], shape)
        off = L * self._layer_stride
        for name, shape in (("embeddings.word_embeddings.weight", (config.vocab_size, H)),
                            ("embeddings.position_embeddings.weight", (config.max_position_embeddings, H)),
                            ("embeddings.token_type_embeddings.weight", (config.type_vocab_size, H)),
                            ("embeddings.LayerNorm.weight", (H,)), ("embeddings.LayerNorm.bias", (H,))):
            layout[name] = (off, shape)
            off += int(np.prod(shape))
        if H % 8 or F % 8:
            raise ValueError("hidden / intermediate sizes must be multiples of 8 (16-byte aligned bf16 rows)")
        self._layout, self._total = layout, off
        flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.names: List[str] = []
        for name, arr in state_dict.items():
            if name.endswith("position_ids") or name.startswith("pooler."):
                continue
            if name not in layout:
                raise KeyError(f"unexpected parameter {name!r}")
            o, shape = layout[name]
            if tuple(arr.shape) != tuple(shape):
                raise ValueError(f"{name}: shape {tuple(arr.shape)} != {tuple(shape)}")
            flat[o : o + int(np.prod(shape))] = torch.from_numpy(np.ascontiguousarray(arr, np.float32)).reshape(-1).to(self.device)
            self.names.append(name)
        missing = set(layout) - set(self.names)
        if missing:
            raise KeyError(f"state dict lacks {sorted(missing)[:3]} ...")
        self._flat = flat
        for name in self.names:   # parameters share the flat buffer's storage
            o, shape = layout[name]
            self.register_parameter(_pname(name), nn.Parameter(flat[o : o + int(np.prod(shape))].view(shape)))
        self._flat_grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self._flat_bf16 = torch.empty(off, dtype=torch.bfloat16, device=self.device)
        self.cfg_struct = _native.GenericConfig(
            config.vocab_size, config.hidden_size, config.num_hidden_layers, config.num_attention_heads,
            config.intermediate_size, config.max_position_embeddings, config.type_vocab_size,
            float(config.layer_norm_eps), pos_offset)
        self._weights_version = -1
        self._seen_versions: Optional[tuple] = None
        self._epoch = 0   # bumped by invalidate(): changes that torch's version counters do not see (HIP-graph replays)
        self._keep: list = []
        self.w_struct = None
        self.g_struct, self._g_layers = self._grad_struct()

    def p(self, hf_name: str) -> nn.Parameter:
        return getattr(self, _pname(hf_name))

    @property
    def flat_grad(self) -> torch.Tensor:
        """every parameter's gradient in ONE fp32 tensor (the buffer all ``p.grad`` are views of): what a data-parallel
        trainer all-reduces after ``backward()`` - the backward writes gradients as side effects, so autograd hooks
        (DDP's mechanism) never fire"""
        return self._flat_grad

    def _check_autograd_contract(self) -> None:
        """The HIP backward hands autograd ``None`` for every parameter and writes ``p.grad`` itself: refuse the uses
        that would silently get no gradient."""
        for name in self.names:
            q = self.p(name)
            if not q.requires_grad:
                raise NotImplementedError(
                    f"{name}: requires_grad=False on a single parameter is not served by the fused backward "
                    "(it writes every gradient); freeze by leaving the parameter out of the optimizer")
            if getattr(q, "_backward_hooks", None) or getattr(q, "_post_accumulate_grad_hooks", None):
                raise NotImplementedError(
                    f"{name} carries autograd hooks (DistributedDataParallel?): the fused backward writes p.grad as a "
                    "side effect and hooks would never fire - all-reduce TrainableEncoder.flat_grad instead")

    def _off(self, layer: int, suffix: str) -> int:
        return layer * self._layer_stride + self._field_off[suffix]

    # -------------------------------------------------------------- device copies
    def invalidate(self) -> None:
        """Force the bf16 copies to be rebuilt at the next forward.  Needed only after parameters were changed through
        ``p.data`` / ``torch.no_grad`` tricks that do not bump the tensors' version counters (in-place optimizer steps
        do bump them and are picked up automatically) - and after REPLAYING a captured training step (GraphedStep): a
        replay updates the parameters on the device without touching torch's counters."""
        self._seen_versions = None
        self._epoch += 1

    def _refresh_device_weights(self) -> None:
        params = list(self.parameters())
        for q in params:   # a parameter re-pointed at foreign storage (load_state_dict copies in place and is fine)
            if q.untyped_storage().data_ptr() != self._flat.untyped_storage().data_ptr():
                raise RuntimeError("a parameter no longer lives in the flat master buffer (assign with p.data.copy_ / load_state_dict)")
        versions = tuple(q._version for q in params)
        if versions == self._seen_versions and self.w_struct is not None:
            return
        cfg = self.config
        H, F, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
        self._flat_bf16.copy_(self._flat)                                   # ONE cast kernel for all parameters
        bf_base, f_base = self._flat_bf16.data_ptr(), self._flat.data_ptr()

        def transposed(suffix: str, rows: int, cols: int) -> torch.Tensor:   # [L, cols, rows], one launch for all layers
            w = self._flat_bf16.as_strided((L, rows, cols), (self._layer_stride, cols, 1), self._field_off[suffix])
            return w.transpose(1, 2).contiguous()

        t_qkv = transposed("attention.self.query.weight", 3 * H, H)
        t_o = transposed("attention.output.dense.weight", H, H)
        t_1 = transposed("intermediate.dense.weight", F, H)
        t_2 = transposed("output.dense.weight", H, F)
        layers = (_native.GenericLayerWeights * max(L, 1))()
        for i in range(L):
            lw = layers[i]
            bf = lambda sfx: bf_base + 2 * self._off(i, sfx)    # noqa: E731
            f32 = lambda sfx: f_base + 4 * self._off(i, sfx)    # noqa: E731
            lw.wqkv, lw.wqkv_t = bf("attention.self.query.weight"), t_qkv[i].data_ptr()
            lw.bqkv = f32("attention.self.query.bias")
            lw.wo, lw.wo_t, lw.bo = bf("attention.output.dense.weight"), t_o[i].data_ptr(), f32("attention.output.dense.bias")
            lw.w1, lw.w1_t, lw.b1 = bf("intermediate.dense.weight"), t_1[i].data_ptr(), f32("intermediate.dense.bias")
            lw.w2, lw.w2_t, lw.b2 = bf("output.dense.weight"), t_2[i].data_ptr(), f32("output.dense.bias")
            lw.ln1_g, lw.ln1_b = f32("attention.output.LayerNorm.weight"), f32("attention.output.LayerNorm.bias")
            lw.ln2_g, lw.ln2_b = f32("output.LayerNorm.weight"), f32("output.LayerNorm.bias")
        w = _native.GenericWeights()
        w.word_emb = bf_base + 2 * self._layout["embeddings.word_embeddings.weight"][0]
        w.pos_emb = bf_base + 2 * self._layout["embeddings.position_embeddings.weight"][0]
        w.type_emb = bf_base + 2 * self._layout["embeddings.token_type_embeddings.weight"][0]
        w.emb_ln_g = f_base + 4 * self._layout["embeddings.LayerNorm.weight"][0]
        w.emb_ln_b = f_base + 4 * self._layout["embeddings.LayerNorm.bias"][0]
        w.layers = layers
        self.w_struct, self._layers_struct, self._keep = w, layers, [t_qkv, t_o, t_1, t_2]
        self._seen_versions = versions
        self._weights_version += 1

    # -------------------------------------------------------------- gradients
    def _grad_struct(self):
        L = self.config.num_hidden_layers
        g_base = self._flat_grad.data_ptr()
        layers = (_native.GenericLayerGrads * max(L, 1))()
        for i in range(L):
            at = lambda sfx: g_base + 4 * self._off(i, sfx)    # noqa: E731
            lg = layers[i]
            lg.wqkv, lg.bqkv = at("attention.self.query.weight"), at("attention.self.query.bias")
            lg.wo, lg.bo = at("attention.output.dense.weight"), at("attention.output.dense.bias")
            lg.ln1_g, lg.ln1_b = at("attention.output.LayerNorm.weight"), at("attention.output.LayerNorm.bias")
            lg.w1, lg.b1 = at("intermediate.dense.weight"), at("intermediate.dense.bias")
            lg.w2, lg.b2 = at("output.dense.weight"), at("output.dense.bias")
            lg.ln2_g, lg.ln2_b = at("output.LayerNorm.weight"), at("output.LayerNorm.bias")
        s = _native.GenericGrads()
        s.word_emb = g_base + 4 * self._layout["embeddings.word_embeddings.weight"][0]
        s.pos_emb = g_base + 4 * self._layout["embeddings.position_embeddings.weight"][0]
        s.type_emb = g_base + 4 * self._layout["embeddings.token_type_embeddings.weight"][0]
        s.emb_ln_g = g_base + 4 * self._layout["embeddings.LayerNorm.weight"][0]
        s.emb_ln_b = g_base + 4 * self._layout["embeddings.LayerNorm.bias"][0]
        s.layers = layers
        return s, layers

    def _attach_grads(self) -> None:
        """Make every ``p.grad`` the matching view of the flat gradient buffer.  Gradients that were ``None`` (after
        ``zero_grad(set_to_none=True)`` or before the first step) start from ONE memset; gradients that already are our
        views keep their contents (autograd semantics: backward accumulates); foreign ``.grad`` tensors are folded in."""
        params = [self.p(n) for n in self.names]
        ours = [q.grad is not None and q.grad.untyped_storage().data_ptr() == self._flat_grad.untyped_storage().data_ptr()
                for q in params]
        if all(ours):
            return
        if not any(ours):
            self._flat_grad.zero_()
        for name, q, mine in zip(self.names, params, ours):
            if mine:
                continue
            o, shape = self._layout[name]
            view = self._flat_grad[o : o + int(np.prod(shape))].view(shape)
            if any(ours):
                view.zero_()                      # mixed state: this slot was dropped, the others are live
            if q.grad is not None:
                view.add_(q.grad)                 # a gradient somebody else put there
            q.grad = view

    # -------------------------------------------------------------- forward
    def forward(self, input_ids, attention_mask=None, normalize: bool = True) -> torch.Tensor:
        """ids / mask ``[B, S]`` -> embeddings ``[B, hidden]`` (fp32, on device, differentiable with
        respect to every parameter).  S is padded to a multiple of 32 here."""
        ids = torch.as_tensor(np.asarray(input_ids) if not isinstance(input_ids, torch.Tensor) else input_ids)
        ids = ids.to(device=self.device, dtype=torch.int32)
        mask = torch.ones_like(ids) if attention_mask is None else torch.as_tensor(
            np.asarray(attention_mask) if not isinstance(attention_mask, torch.Tensor) else attention_mask
        ).to(device=self.device, dtype=torch.int32)
        B, S = ids.shape
        Sp = max(32, -(-S // 32) * 32)
        if Sp != S:
            ids = torch.nn.functional.pad(ids, (0, Sp - S))
            mask = torch.nn.functional.pad(mask, (0, Sp - S))
        params = [self.p(n) for n in self.names]
        return _EncoderFunction.apply(self, ids.contiguous(), mask.contiguous(), normalize, *params)

    def state_dict_numpy(self) -> Dict[str, np.ndarray]:
        return {n: self.p(n).detach().cpu().numpy() for n in self.names}


def kd_step_scores(query_emb: torch.Tensor, doc_embs: torch.Tensor) -> torch.Tensor:
    """``torch.matmul(query_emb, doc_embs.T)[0]`` of the reference step (src/kd/train.py:189)."""
    return torch.matmul(query_emb, doc_embs.T)[0]


class GraphedStep:
    """ONE training step - every encoder forward, the loss, the backward pass and the optimizer update - captured into
    a HIP graph and replayed (reference step: src/kd/train.py:176-210).

    The KD step is ~800 short launches per 30 ms: replayed as one graph it no longer depends on how fast the host can
    issue them (round 3 measured 111-132 ms for the same 30 ms of kernels on a box whose host cores were busy).
    Nothing in the C-ABI allocates or synchronises, so every launch of a step is capturable; torch's caching allocator
    serves the step's temporaries from the graph's private pool.

    ``step_fn()`` runs one whole step on the CURRENT stream and returns a tensor (the loss) or a tuple of tensors; it
    must read its batch from tensors that keep their storage (``GraphedStep.copy_inputs`` / ``tensor.copy_``) and must
    not synchronise (no ``.item()``, no host reads).  The optimizer must be created with ``capturable=True``.
    ``opt.zero_grad(set_to_none=True)`` inside ``step_fn`` is fine: the gradient views are re-attached during capture
    and the memset of the flat gradient buffer is part of the graph."""

    def __init__(self, step_fn, warmup: int = 3, modules: Sequence["TrainableEncoder"] = ()) -> None:
        """``modules``: the ``TrainableEncoder``s the step trains.  A replay changes their parameters without bumping
        torch's version counters, so every replay marks them stale (``invalidate()``): the next EAGER forward - an
        evaluation between steps, ``Mi355xSentenceEncoder.encode`` - then re-casts the weights instead of using the bf16
        copies of one optimizer step ago."""
        _native.require_gpu()
        self._fn = step_fn
        self._modules = list(modules)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up off the default stream, as torch's capture rules ask
            for _ in range(max(int(warmup), 1)):
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = step_fn()
        self.steps_replayed = 0

    def __call__(self):
        self.graph.replay()
        self.steps_replayed += 1
        for m in self._modules:
            m.invalidate()
        return self.out

    @staticmethod
    def copy_inputs(static: Sequence[torch.Tensor], fresh: Sequence[torch.Tensor]) -> None:
        """copy the next batch into the tensors the captured step reads (same shapes)"""
        for dst, src in zip(static, fresh):
            if dst.shape != src.shape:
                raise ValueError(f"a captured step has fixed shapes: {tuple(dst.shape)} != {tuple(src.shape)}")
            dst.copy_(src, non_blocking=True)
