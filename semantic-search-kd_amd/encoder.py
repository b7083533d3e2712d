"""``SentenceTransformer.encode()``-shaped bi-encoder running on the gfx950 HIP kernels.

This is the object the reference's ``StudentModel`` holds as ``.model`` (reference:
src/kd/train.py:126-127, tests/test_student_model.py:12-70): constructed from a model path and a
device, exposing ``encode(...)``, ``get_sentence_embedding_dimension()`` and
``max_seq_length``.  The pipeline is the e5-small-v2 one — Transformer -> Pooling(mean) ->
Normalize (tests/test_model_validation.py:80-89,256-262) — executed by
``sskd_encoder_forward`` (bf16 MFMA encoder + fused mean-pool / L2-normalise).
There is no CPU path: without an MI355X and the built library, construction fails.
"""
from __future__ import annotations

import ctypes
import itertools
import json
import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _native
from .weights import BertConfig, DeviceWeights, load_config, load_state_dict, synthetic_state_dict

CLS_ID, SEP_ID, PAD_ID = 101, 102, 0

ROW_CAPACITY = 256          # tokens per packed row (8 token tiles: one attention workgroup)
LAUNCH_TOKENS = 512 * 256   # padded tokens per forward launch (the benchmarked batch 512 x 256 shape)
TOKENIZE_CHUNK = 8192       # texts tokenised per background task while the GPU encodes the previous chunk


class _Staging:
    """Pinned host buffer + device twin for one launch's inputs (flat ids | cu_seqlens | plan table),
    moved with ONE asynchronous copy.  A small ring of these lets the host fill launch i + 1 while
    the copy of launch i is still in flight."""

    def __init__(self, device, words: int):
        self.host = torch.empty(words, dtype=torch.int32).pin_memory()
        self.dev = torch.empty(words, dtype=torch.int32, device=device)
        self.np = self.host.numpy()
        self.done = torch.cuda.Event()
        self.used = False


class NativeWordPiece:
    """The C++ WordPiece tokenizer of the C-ABI library (``sskd_tokenizer_*``), built from a
    ``tokenizers.Tokenizer`` when - and only when - that tokenizer is the uncased BERT recipe it
    implements (BertNormalizer lowercase, BertPreTokenizer, WordPiece "##", [CLS] $A [SEP])."""

    def __init__(self, vocab_tokens: Sequence[str], max_chars_per_word: int = 100):
        lib = _native.load()
        blob = "\n".join(vocab_tokens).encode("utf-8")
        self._handle = ctypes.c_void_p()
        _native.check(lib.sskd_tokenizer_create(blob, len(blob), ctypes.byref(self._handle)))
        self._lib = lib
        self.threads = max(1, min(16, os.cpu_count() or 1))
        if max_chars_per_word != 100:
            raise ValueError("the native tokenizer implements max_input_chars_per_word = 100")

    @classmethod
    def from_hf(cls, tok) -> Optional["NativeWordPiece"]:
        try:
            spec = json.loads(tok.to_str())
            model, norm = spec.get("model") or {}, spec.get("normalizer") or {}
            pre, post = spec.get("pre_tokenizer") or {}, spec.get("post_processor") or {}
            ok = (
                model.get("type") == "WordPiece"
                and model.get("continuing_subword_prefix", "##") == "##"
                and model.get("unk_token") == "[UNK]"
                and model.get("max_input_chars_per_word", 100) == 100
                and norm.get("type") == "BertNormalizer"
                and norm.get("lowercase", True) is True
                and norm.get("clean_text", True) is True
                and pre.get("type") == "BertPreTokenizer"
                and post.get("type") in ("TemplateProcessing", "BertProcessing")
            )
            if not ok:
                return None
            vocab = model["vocab"]
            toks = [None] * len(vocab)
            for t, i in vocab.items():
                if not 0 <= i < len(toks) or toks[i] is not None or "\n" in t:
                    return None
                toks[i] = t
            self = cls(toks)
            # the library matches ADDED tokens ([CLS], [SEP], [MASK], ... and any user-added ones) in the RAW text,
            # before normalisation, and emits their single id; the C++ path would split "[SEP]" into "[", "sep", "]".
            # Texts containing one of these strings are therefore routed to the library (as non-ASCII texts are).
            self.added_tokens = tuple(sorted({a["content"] for a in spec.get("added_tokens") or [] if a.get("content")}))
            return self
        except Exception:
            return None

    def needs_library(self, text: str) -> bool:
        """True when ``text`` contains a literal added-token string (only the library tokenizes those as one id)."""
        return any(t in text for t in getattr(self, "added_tokens", ()))

    def encode_flat(self, texts: Sequence[str], max_len: int):
        """-> (flat int32 ids, lengths int32, needs_unicode bool[n]) or None when the texts cannot be
        NUL-joined (a text contains NUL)."""
        n = len(texts)
        blob = "\x00".join(texts).encode("utf-8")
        seps = np.flatnonzero(np.frombuffer(blob, np.uint8) == 0)
        if seps.size != max(n - 1, 0):
            return None
        offsets = np.empty(n + 1, np.int64)
        offsets[0] = 0
        offsets[1:n] = seps + 1
        offsets[n] = len(blob)
        cap = len(blob) + 2 * n + 16  # every id consumes at least one input byte, plus [CLS] / [SEP]
        ids = np.empty(cap, np.int32)
        lengths = np.empty(n, np.int32)
        flags = np.empty(n, np.uint8)
        total = ctypes.c_int64(0)
        _native.check(self._lib.sskd_tokenizer_encode(
            self._handle, blob, offsets.ctypes.data, n, max_len, self.threads, ids.ctypes.data, cap,
            lengths.ctypes.data, flags.ctypes.data, ctypes.byref(total)))
        return ids[: total.value], lengths, flags.astype(bool)

    def __del__(self):
        try:
            if self._handle:
                self._lib.sskd_tokenizer_destroy(self._handle)
        except Exception:
            pass


class Mi355xSentenceEncoder:
    def __init__(
        self,
        model_name_or_path: Union[str, Path, None] = None,
        device: Optional[str] = None,
        *,
        config: Optional[BertConfig] = None,
        state_dict: Optional[Dict[str, np.ndarray]] = None,
        tokenizer=None,
        max_seq_length: Optional[int] = None,
    ) -> None:
        """Load weights from a LOCAL HF / sentence-transformers directory, or take an explicit
        ``config`` + ``state_dict`` (synthetic weights).  Model *names* are not fetched."""
        _native.require_gpu()
        self.device = _resolve_device(device)
        model_dir = None
        if state_dict is None:
            if model_name_or_path is None:
                raise ValueError("pass a local model directory or config= / state_dict=")
            model_dir = Path(model_name_or_path)
            if not model_dir.is_dir():
                raise FileNotFoundError(
                    f"{model_name_or_path!r} is not a local directory. This backend never downloads "
                    "checkpoints: point it at a directory holding config.json + model.safetensors + tokenizer.json"
                )
            config = load_config(model_dir)
            state_dict = load_state_dict(model_dir)
        self.config = config or BertConfig()
        self.weights = DeviceWeights(self.config, state_dict, self.device)
        self._host_state = state_dict      # fp32 masters, the seed of the trainable copy
        self._trainable = None             # training.TrainableEncoder, created on first use
        self._trainable_versions = None
        self._tokenizer = None
        self._native_tok: Optional[NativeWordPiece] = None
        self.tokenizer = tokenizer if tokenizer is not None else _load_tokenizer(model_dir)
        st_max = _read_st_max_len(model_dir)
        self.max_seq_length = int(
            max_seq_length or st_max or min(512, self.config.max_position_embeddings)
        )
        self._workspace: Optional[torch.Tensor] = None
        self._workspace2: Optional[torch.Tensor] = None   # second half of a split batch (side stream)
        self._side_stream: Optional[torch.cuda.Stream] = None
        self._copy_stream: Optional[torch.cuda.Stream] = None
        # large batches / alternate launches over two HIP streams: gave 4 % while the output projection
        # was a separate load/store-bound kernel; with it fused into the MLP prologue it is neutral (off)
        self.split_streams = False
        self._staging: List[_Staging] = []
        self._stage_next = 0
        self._rows: list = [None, None]   # per stream lane: (packed ids, segment words)
        self._tok_pool: Optional[ThreadPoolExecutor] = None
        self.last_encode_stats: Dict[str, float] = {}

    @property
    def tokenizer(self):
        return self._tokenizer

    @tokenizer.setter
    def tokenizer(self, tok) -> None:
        """The ``tokenizers.Tokenizer`` (always the reference for text -> ids); when it is the uncased
        BERT recipe, ASCII texts are tokenised by the multi-threaded C++ twin instead."""
        self._tokenizer = tok
        self._native_tok = NativeWordPiece.from_hf(tok) if tok is not None else None

    # ----------------------------------------------------------- constructors
    @classmethod
    def from_synthetic(
        cls, config: Optional[BertConfig] = None, device: Optional[str] = None, tokenizer=None,
        stress: bool = False, **kw
    ) -> "Mi355xSentenceEncoder":
        """Random-init weights of the e5-small-v2 architecture (deterministic recipe, weights.py);
        ``stress`` = the trained-checkpoint-like hard-case recipe (peaky attention, outlier channels)."""
        cfg = config or BertConfig()
        sd = synthetic_state_dict(cfg, stress=stress)
        return cls(None, device, config=cfg, state_dict=sd, tokenizer=tokenizer, **kw)

    # ------------------------------------------------------ training surface (src/kd/train.py:126-127,154)
    def trainable(self):
        """The fp32 master parameters as an ``nn.Module`` whose forward/backward run in HIP
        (``training.TrainableEncoder``); created on first use from the loaded weights."""
        if self._trainable is None:
            from .training import TrainableEncoder

            self._trainable = TrainableEncoder(self.config, self._host_state, self.device)
            self._trainable_versions = (self._trainable._epoch,) + tuple(p._version for p in self._trainable.parameters())
        return self._trainable

    def parameters(self):
        return self.trainable().parameters()

    def named_parameters(self):
        return self.trainable().named_parameters()

    def train(self, mode: bool = True):
        self.trainable().train(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        if _resolve_device(str(device)) != self.device:
            raise RuntimeError(f"the encoder lives on {self.device}; re-create it on {device}")
        return self

    def sync_inference_weights(self) -> bool:
        """Re-tile the bf16 inference weights from the trained fp32 masters when they changed
        (called by ``encode`` so that evaluation after optimizer steps sees the new weights)."""
        if self._trainable is None:
            return False
        versions = (self._trainable._epoch,) + tuple(p._version for p in self._trainable.parameters())
        if versions == self._trainable_versions:
            return False
        self._host_state = self._trainable.state_dict_numpy()
        self.weights = DeviceWeights(self.config, self._host_state, self.device)
        self._trainable_versions = versions
        return True

    # ------------------------------------------------------ SentenceTransformer API
    def get_sentence_embedding_dimension(self) -> int:
        return self.config.hidden_size

    def tokenize(self, texts: Sequence[str]) -> Dict[str, np.ndarray]:
        """WordPiece ids, padded to the longest text, truncated to ``max_seq_length``."""
        if self.tokenizer is None:
            raise RuntimeError(
                "no tokenizer: the model directory has neither tokenizer.json nor vocab.txt "
                "(use encode_token_ids for pre-tokenised input)"
            )
        encs = self.tokenizer.encode_batch(list(texts))
        rows = [e.ids[: self.max_seq_length] for e in encs]
        for r, e in zip(rows, encs):
            if len(e.ids) > self.max_seq_length and r:
                r[-1] = e.ids[-1]  # keep the closing [SEP] when truncating
        width = max((len(r) for r in rows), default=1) or 1
        ids = np.full((len(rows), width), PAD_ID, np.int32)
        mask = np.zeros((len(rows), width), np.int32)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = r
            mask[i, : len(r)] = 1
        return {"input_ids": ids, "attention_mask": mask}

    def encode_token_ids(
        self, input_ids, attention_mask=None, normalize: bool = True, out: Optional[torch.Tensor] = None
    ) -> torch.Tensor:
        """One forward pass over pre-tokenised ``[B, S]`` int32 ids; returns fp32 ``[B, 384]`` on device.

        Everything is enqueued on the current stream; no host synchronisation.
        """
        lib = _native.load()
        self.sync_inference_weights()
        ids = _as_device_i32(input_ids, self.device)
        if ids.dim() != 2:
            raise ValueError(f"input_ids must be [B, S], got {tuple(ids.shape)}")
        B, S = ids.shape
        mask = torch.ones_like(ids) if attention_mask is None else _as_device_i32(attention_mask, self.device)
        if mask.shape != ids.shape:
            raise ValueError("attention_mask shape differs from input_ids")
        if out is None:
            out = torch.empty((B, self.config.hidden_size), dtype=torch.float32, device=self.device)
        if B == 0:
            return out
        if self.split_streams and B >= 512 and B % 2 == 0 and S <= 256:
            # Two halves on two HIP streams: the load/store-bound output projection of one half runs
            # beside the matrix-bound fused MLP of the other (same work, 5-6 % less time at 512 x 256).
            # Fork / join with events: the caller's stream semantics are unchanged.
            h = B // 2
            main = torch.cuda.current_stream(self.device)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(self.device)
            side = self._side_stream
            side.wait_stream(main)
            self._forward_rows(lib, ids[:h], mask[:h], normalize, out[:h], main, "_workspace")
            with torch.cuda.stream(side):
                self._forward_rows(lib, ids[h:], mask[h:], normalize, out[h:], side, "_workspace2")
            main.wait_stream(side)
            return out
        self._forward_rows(lib, ids, mask, normalize, out, torch.cuda.current_stream(self.device), "_workspace")
        return out

    def capture_forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                        normalize: bool = True, out: Optional[torch.Tensor] = None) -> "GraphedForward":
        """The forward over a FIXED ``[B, S]`` shape captured into one HIP graph (26 launches at 12 layers): the caller
        keeps refilling ``input_ids`` / ``attention_mask`` (device int32 tensors, same storage) and calls ``replay()``;
        the embeddings land in ``.out``.  For loops that encode many equally shaped batches (index builds at a fixed
        token budget, the bench): the launches no longer depend on the host's pace."""
        ids = _as_device_i32(input_ids, self.device)
        mask = torch.ones_like(ids) if attention_mask is None else _as_device_i32(attention_mask, self.device)
        if ids.data_ptr() != (input_ids.data_ptr() if isinstance(input_ids, torch.Tensor) else 0):
            raise ValueError("capture_forward needs device int32 tensors (the graph reads THEIR storage at every replay)")
        if out is None:
            out = torch.empty((ids.shape[0], self.config.hidden_size), dtype=torch.float32, device=self.device)
        return GraphedForward(self, ids, mask, bool(normalize), out)

    def _forward_rows(self, lib, ids, mask, normalize, out, stream, ws_name: str) -> None:
        B, S = ids.shape
        need = int(lib.sskd_encoder_workspace_bytes(self.weights.cstruct_cfg, B, S))
        ws = getattr(self, ws_name)
        if ws is None or ws.numel() < need:
            setattr(self, ws_name, None)
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            setattr(self, ws_name, ws)
        _native.check(
            lib.sskd_encoder_forward(
                self.weights.cstruct_cfg,
                self.weights.struct,
                ids.data_ptr(),
                mask.data_ptr(),
                B,
                S,
                int(bool(normalize)),
                out.data_ptr(),
                ws.data_ptr(),
                ws.numel(),
                int(stream.cuda_stream),
            )
        )

    # ------------------------------------------------------------ packed varlen path
    def _stage(self, words: int) -> _Staging:
        if not self._staging or self._staging[0].host.numel() < words:
            size = max(words, LAUNCH_TOKENS + 5 * (LAUNCH_TOKENS // 8) + 64)
            self._staging = [_Staging(self.device, size) for _ in range(4)]
            self._stage_next = 0
        st = self._staging[self._stage_next]
        self._stage_next = (self._stage_next + 1) % len(self._staging)
        if st.used:
            st.done.synchronize()  # its previous copy has been consumed by the GPU
        return st

    def encode_ragged(self, flat_ids: np.ndarray, lengths: np.ndarray, normalize: bool = True,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Varlen forward: ``flat_ids`` holds the sequences back to back (int32), ``lengths[i]`` tokens each
        (1 .. ROW_CAPACITY).  Whole sequences are packed into rows of ROW_CAPACITY tokens
        (``sskd_pack_plan``: best-fit decreasing, so only row tails are padding), attention is
        block-diagonal per sequence, and the pooled embedding of sequence i lands in ``out[i]``:
        no length sorting, no per-batch padding, no scatter.  Launches are cut by a token budget
        (LAUNCH_TOKENS), not by a caller batch size.  Enqueued on the current stream."""
        lib = _native.load()
        self.sync_inference_weights()
        lengths = np.ascontiguousarray(lengths, np.int32)
        flat_ids = np.ascontiguousarray(flat_ids, np.int32)
        n = int(lengths.shape[0])
        if out is None:
            out = torch.empty((n, self.config.hidden_size), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        if int(lengths.min()) < 1 or int(lengths.max()) > ROW_CAPACITY:
            raise ValueError(f"encode_ragged: lengths must lie in [1, {ROW_CAPACITY}]")
        cu = np.zeros(n + 1, np.int64)
        np.cumsum(lengths, out=cu[1:])
        if int(cu[-1]) != flat_ids.shape[0]:
            raise ValueError("encode_ragged: flat_ids does not hold sum(lengths) tokens")
        # launches alternate between the caller's stream and a side stream (fork / join with events):
        # the load/store-bound kernels of one launch run beside the matrix-bound ones of the other
        main = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(self.device)
        lanes = [main, self._side_stream if self.split_streams else main]
        self._side_stream.wait_stream(main)
        budget = int(LAUNCH_TOKENS * 0.97)
        padded_tokens = 0
        s0 = 0
        launch = 0
        while s0 < n:
            s1 = int(np.searchsorted(cu, cu[s0] + budget, side="right")) - 1
            s1 = min(max(s1, s0 + 1), n)
            m, t0, t1 = s1 - s0, int(cu[s0]), int(cu[s1])
            total = t1 - t0
            cap = ROW_CAPACITY if total >= ROW_CAPACITY else -(-total // 32) * 32
            # staging layout (int32 words): [tokens | cu_seqlens (m + 1) | pad to 4 words | table (4 per sequence)]; the
            # kernels read table entries as 16-byte vectors, so the table starts on a 16-byte boundary
            tab0 = -(-(total + m + 1) // 4) * 4
            st = self._stage(tab0 + 4 * m)
            a = st.np
            a[:total] = flat_ids[t0:t1]
            a[total : total + m + 1] = cu[s0 : s1 + 1] - t0
            table = a[tab0 : tab0 + 4 * m]
            n_rows = ctypes.c_int(0)
            _native.check(lib.sskd_pack_plan(lengths[s0:s1].ctypes.data, m, cap, table.ctypes.data, n_rows))
            rows = n_rows.value
            words = tab0 + 4 * m
            lane = launch & 1
            with torch.cuda.stream(lanes[lane]):
                stream = int(lanes[lane].cuda_stream)
                st.dev[:words].copy_(st.host[:words], non_blocking=True)
                st.done.record(lanes[lane])
                st.used = True
                need_rows = rows * cap
                if self._rows[lane] is None or self._rows[lane][0].numel() < need_rows:
                    size = max(need_rows, LAUNCH_TOKENS + 16 * ROW_CAPACITY)
                    self._rows[lane] = (torch.empty(size, dtype=torch.int32, device=self.device),
                                        torch.empty(size, dtype=torch.int32, device=self.device))
                rows_ids, rows_seg = self._rows[lane]
                base = st.dev.data_ptr()
                d_cu, d_table = base + 4 * total, base + 4 * tab0
                assert d_table % 16 == 0
                _native.check(lib.sskd_pack_tokens(base, d_cu, d_table, m, rows, cap, rows_ids.data_ptr(),
                                                   rows_seg.data_ptr(), stream))
                need = int(lib.sskd_encoder_workspace_bytes(self.weights.cstruct_cfg, rows, cap))
                ws_name = "_workspace" if lane == 0 else "_workspace2"
                ws = getattr(self, ws_name)
                if ws is None or ws.numel() < need:
                    setattr(self, ws_name, None)
                    ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                    setattr(self, ws_name, ws)
                _native.check(
                    lib.sskd_encoder_forward_packed(
                        self.weights.cstruct_cfg, self.weights.struct, rows_ids.data_ptr(), rows_seg.data_ptr(),
                        rows, cap, d_table, m, int(bool(normalize)), out[s0:].data_ptr(),
                        ws.data_ptr(), ws.numel(), stream,
                    )
                )
            padded_tokens += rows * cap
            s0 = s1
            launch += 1
        main.wait_stream(self._side_stream)
        self.last_encode_stats = {"real_tokens": float(cu[-1]), "padded_tokens": float(padded_tokens),
                                  "padding_overhead": padded_tokens / float(cu[-1]) - 1.0}
        return out

    def _tokenize_flat(self, texts: Sequence[str]):
        """WordPiece ids of ``texts`` as one flat int32 stream + lengths (truncated to max_seq_length,
        keeping the closing [SEP])."""
        if self.tokenizer is None:
            raise RuntimeError(
                "no tokenizer: the model directory has neither tokenizer.json nor vocab.txt "
                "(use encode_token_ids / encode_ragged for pre-tokenised input)"
            )
        mx = self.max_seq_length
        fast = getattr(self.tokenizer, "encode_batch_fast", None) or self.tokenizer.encode_batch
        if self._native_tok is not None and len(texts):
            res = self._native_tok.encode_flat(texts, mx)
            if res is not None:
                flat, lengths, uni = res
                if getattr(self._native_tok, "added_tokens", ()):
                    # cheap pre-filter: every BERT added token starts with '[' - only such texts are scanned further
                    for j, t in enumerate(texts):
                        if "[" in t and self._native_tok.needs_library(t):
                            uni[j] = True
                if not uni.any():
                    return flat, lengths
                # texts with non-ASCII characters: the Unicode-complete tokenizer, spliced back in place
                idx = np.flatnonzero(uni)
                pieces = np.split(flat, np.cumsum(lengths)[:-1])
                for i, e in zip(idx, fast([texts[j] for j in idx])):
                    r = e.ids if len(e.ids) <= mx else e.ids[: mx - 1] + e.ids[-1:]
                    pieces[i] = np.asarray(r or [PAD_ID], np.int32)
                    lengths[i] = len(pieces[i])
                return np.concatenate(pieces), lengths
        encs = fast(list(texts))
        rows = [e.ids for e in encs]
        lengths = np.fromiter((len(r) for r in rows), np.int32, count=len(rows))
        if len(rows) and int(lengths.max()) > mx:
            rows = [r if len(r) <= mx else r[: mx - 1] + r[-1:] for r in rows]
            lengths = np.minimum(lengths, mx)
        if len(rows) and int(lengths.min()) < 1:  # a tokenizer without a [CLS] .. [SEP] template
            rows = [r if r else [PAD_ID] for r in rows]
            lengths = np.maximum(lengths, 1)
        flat = np.fromiter(itertools.chain.from_iterable(rows), np.int32, count=int(lengths.sum()))
        return flat, lengths

    def hidden_states(self, input_ids, attention_mask=None) -> torch.Tensor:
        """bf16 ``[B, S, 384]`` output of the last encoder layer (test hook)."""
        lib = _native.load()
        ids = _as_device_i32(input_ids, self.device)
        B, S = ids.shape
        mask = torch.ones_like(ids) if attention_mask is None else _as_device_i32(attention_mask, self.device)
        out = torch.empty((B, S, self.config.hidden_size), dtype=torch.bfloat16, device=self.device)
        need = int(lib.sskd_encoder_workspace_bytes(self.weights.cstruct_cfg, B, S))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        _native.check(
            lib.sskd_encoder_hidden(
                self.weights.cstruct_cfg, self.weights.struct, ids.data_ptr(), mask.data_ptr(), B, S,
                out.data_ptr(), ws.data_ptr(), ws.numel(), int(torch.cuda.current_stream(self.device).cuda_stream),
            )
        )
        return out

    def encode(
        self,
        sentences: Union[str, List[str]],
        batch_size: int = 32,
        show_progress_bar: bool = False,
        convert_to_numpy: bool = True,
        convert_to_tensor: bool = False,
        normalize_embeddings: bool = False,
        device: Optional[str] = None,
        **_ignored,
    ):
        """Same contract as ``SentenceTransformer.encode`` (str or list in, [n, 384] out, input order
        kept); the e5 pipeline ends in a ``Normalize`` module, so embeddings are unit-norm whatever
        ``normalize_embeddings`` says.  Where sentence-transformers sorts by length and pads every
        batch of ``batch_size`` texts to its longest member, this path packs whole sequences into
        256-token rows (``encode_ragged``) and cuts launches by a token budget: ``batch_size`` (the
        reference CLI default is 32) does not shape the GPU work.  Texts are tokenised in chunks on a
        background thread (the Rust tokenizer releases the GIL) while the GPU encodes the previous
        chunk.  Texts longer than one row (> 256 tokens) take the padded ``encode_token_ids`` path."""
        del show_progress_bar, device, normalize_embeddings, batch_size
        single = isinstance(sentences, str)
        texts = [sentences] if single else list(sentences)
        n = len(texts)
        out = torch.empty((n, self.config.hidden_size), dtype=torch.float32, device=self.device)
        # NumPy out: every chunk's rows start their way to pinned host memory as soon as the chunk is
        # enqueued, so the device-to-host copy hides behind the next chunk's forward
        to_host = convert_to_numpy or not convert_to_tensor
        host = torch.empty((n, self.config.hidden_size), dtype=torch.float32, pin_memory=True) if (to_host and n) else None
        real = padded = 0.0
        if n:
            if self._tok_pool is None:
                self._tok_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="sskd-tokenize")
            bounds = list(range(0, n, TOKENIZE_CHUNK)) + [n]
            pending = self._tok_pool.submit(self._tokenize_flat, texts[bounds[0] : bounds[1]])
            with torch.cuda.device(self.device):
                for ci in range(len(bounds) - 1):
                    flat, lengths = pending.result()
                    if ci + 2 < len(bounds):
                        pending = self._tok_pool.submit(self._tokenize_flat, texts[bounds[ci + 1] : bounds[ci + 2]])
                    lo = bounds[ci]
                    long_rows = np.nonzero(lengths > ROW_CAPACITY)[0]
                    if long_rows.size == 0:
                        self.encode_ragged(flat, lengths, normalize=True, out=out[lo : bounds[ci + 1]])
                        real += self.last_encode_stats["real_tokens"]
                        padded += self.last_encode_stats["padded_tokens"]
                        if host is not None:
                            self._rows_to_host(out, host, lo, bounds[ci + 1])
                        continue
                    cu = np.zeros(len(lengths) + 1, np.int64)
                    np.cumsum(lengths, out=cu[1:])
                    short = np.nonzero(lengths <= ROW_CAPACITY)[0]
                    if short.size:
                        keep = np.concatenate([flat[cu[i] : cu[i + 1]] for i in short])
                        emb = self.encode_ragged(keep, lengths[short], normalize=True)
                        out[torch.from_numpy(lo + short).to(self.device)] = emb
                        real += self.last_encode_stats["real_tokens"]
                        padded += self.last_encode_stats["padded_tokens"]
                    for i0 in range(0, long_rows.size, 64):
                        idx = long_rows[i0 : i0 + 64]
                        width = int(lengths[idx].max())
                        ids = np.full((idx.size, width), PAD_ID, np.int32)
                        mask = np.zeros((idx.size, width), np.int32)
                        for j, i in enumerate(idx):
                            ids[j, : lengths[i]] = flat[cu[i] : cu[i + 1]]
                            mask[j, : lengths[i]] = 1
                        out[torch.from_numpy(lo + idx).to(self.device)] = self.encode_token_ids(ids, mask, normalize=True)
                        real += float(lengths[idx].sum())
                        padded += float(idx.size * (-(-width // 32) * 32))
                    if host is not None:
                        self._rows_to_host(out, host, lo, bounds[ci + 1])
        self.last_encode_stats = {"real_tokens": real, "padded_tokens": padded,
                                  "padding_overhead": (padded / real - 1.0) if real else 0.0}
        if not to_host:
            return out[0] if single else out
        if host is None:
            arr = np.empty((0, self.config.hidden_size), np.float32)
        else:
            self._copy_stream.synchronize()
            arr = host.numpy()   # a view of the pinned block: no second copy; freed with the array
        return arr[0] if single else arr

    def _rows_to_host(self, out: torch.Tensor, host: torch.Tensor, lo: int, hi: int) -> None:
        """Start the device-to-host copy of finished rows on a copy stream of its own (ordered after the
        work enqueued so far), so the next chunk's kernels do not queue behind it."""
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(self.device)
        self._copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._copy_stream):
            host[lo:hi].copy_(out[lo:hi], non_blocking=True)
        out.record_stream(self._copy_stream)

    def cleanup(self) -> None:
        self._workspace = None
        self._workspace2 = None
        self._staging, self._rows = [], [None, None]
        if self._tok_pool is not None:
            self._tok_pool.shutdown(wait=True)
            self._tok_pool = None


# ---------------------------------------------------------------------- helpers
def _resolve_device(device: Optional[str]) -> torch.device:
    if device is None or device == "cuda":
        return torch.device("cuda", torch.cuda.current_device())
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"device={device!r}: the encoder runs on MI355X only (PyTorch-ROCm spells it 'cuda[:N]'); no CPU path"
        )
    return dev


def _as_device_i32(x, device: torch.device) -> torch.Tensor:
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    return t.to(device=device, dtype=torch.int32).contiguous()


def _load_tokenizer(model_dir: Optional[Path]):
    if model_dir is None:
        return None
    from tokenizers import Tokenizer

    tj = model_dir / "tokenizer.json"
    if tj.exists():
        tok = Tokenizer.from_file(str(tj))
        tok.no_padding()
        tok.no_truncation()
        return tok
    vocab = model_dir / "vocab.txt"
    if vocab.exists():
        return build_wordpiece_tokenizer(vocab.read_text().splitlines())
    return None


def build_wordpiece_tokenizer(vocab_tokens: Sequence[str]):
    """Uncased BERT WordPiece tokenizer over an explicit vocabulary (``[CLS] x [SEP]`` template)."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors

    vocab = {t: i for i, t in enumerate(vocab_tokens)}
    tok = Tokenizer(models.WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
    tok.normalizer = normalizers.BertNormalizer(lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.post_processor = processors.TemplateProcessing(
        single="[CLS] $A [SEP]",
        pair="[CLS] $A [SEP] $B:1 [SEP]:1",
        special_tokens=[("[CLS]", vocab["[CLS]"]), ("[SEP]", vocab["[SEP]"])],
    )
    return tok


def _read_st_max_len(model_dir: Optional[Path]) -> Optional[int]:
    if model_dir is None:
        return None
    p = model_dir / "sentence_bert_config.json"
    if p.exists():
        try:
            return int(json.loads(p.read_text()).get("max_seq_length"))
        except Exception:
            return None
    return None


class GraphedForward:
    """``Mi355xSentenceEncoder.capture_forward``: one captured forward over fixed-shape inputs."""

    def __init__(self, enc: "Mi355xSentenceEncoder", ids: torch.Tensor, mask: torch.Tensor, normalize: bool, out: torch.Tensor) -> None:
        self.enc, self.ids, self.mask, self.normalize, self.out = enc, ids, mask, normalize, out
        enc.sync_inference_weights()
        self._weights = enc.weights
        side = torch.cuda.Stream(enc.device)
        side.wait_stream(torch.cuda.current_stream(enc.device))
        with torch.cuda.stream(side):          # warm-up off the capture stream: workspace allocation, code load
            enc.encode_token_ids(ids, mask, normalize=normalize, out=out)
        torch.cuda.current_stream(enc.device).wait_stream(side)
        torch.cuda.synchronize(enc.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            enc.encode_token_ids(ids, mask, normalize=normalize, out=out)

    def replay(self) -> torch.Tensor:
        if self.enc.weights is not self._weights or self.enc.sync_inference_weights():
            raise RuntimeError("the encoder's weights changed since this forward was captured: capture it again")
        self.graph.replay()
        return self.out
