"""Encoder legs of ``bench.py`` and ``__graft_entry__.smoke()`` (product code only: no oracle here)."""
from __future__ import annotations

import time

import numpy as np
import torch

from .encoder import Mi355xSentenceEncoder
from .weights import BertConfig

MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16


def encoder_flops(tokens: int, seq_len: int, cfg: BertConfig) -> float:
    """Algorithmic FLOPs of the forward pass over real tokens:
    per token 12 * (2 * (4 H^2 + 2 H F) + 4 S H)."""
    h, f = cfg.hidden_size, cfg.intermediate_size
    return tokens * cfg.num_hidden_layers * (2.0 * (4 * h * h + 2 * h * f) + 4.0 * seq_len * h)


def synthetic_ids(batch: int, seq_len: int, vocab: int, device, seed: int = 0):
    """ids uniform in [999, vocab), [CLS] first, [SEP] last, full mask."""
    g = torch.Generator(device=device).manual_seed(seed)
    ids = torch.randint(999, vocab, (batch, seq_len), generator=g, device=device, dtype=torch.int32)
    ids[:, 0] = 101
    ids[:, -1] = 102
    return ids, torch.ones_like(ids)


def bench_encode(device, world: int, steps: int, warmup: int, barrier, batch: int = 512, seq_len: int = 256,
                 ragged: bool = True, text: bool = False):
    """docs embedded / s: every rank encodes its own ``batch x seq_len`` synthetic batches
    (pure data parallel, no communication); whole-job rate = world * batch * steps / max-rank time."""
    import torch.distributed as dist

    cfg = BertConfig()
    enc = Mi355xSentenceEncoder.from_synthetic(cfg, device=str(device))
    ids, mask = synthetic_ids(batch, seq_len, cfg.vocab_size, device, seed=int(device.index or 0))
    out = torch.empty((batch, cfg.hidden_size), dtype=torch.float32, device=device)
    for _ in range(max(warmup, 1)):
        enc.encode_token_ids(ids, mask, normalize=True, out=out)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        enc.encode_token_ids(ids, mask, normalize=True, out=out)
    barrier()
    dt_eager = time.perf_counter() - t0
    # the same forward captured into one HIP graph (Mi355xSentenceEncoder.capture_forward): all 26 launches of a step
    # are inside the graph and inside the timed region
    graph_note = None
    try:
        fwd = enc.capture_forward(ids, mask, normalize=True, out=out)
        for _ in range(max(warmup, 1)):
            fwd.replay()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fwd.replay()
        barrier()
        dt = time.perf_counter() - t0
    except Exception as exc:  # noqa: BLE001 - reported in the line; the eager time stands
        graph_note = f"graph capture failed ({type(exc).__name__}: {exc}); eager launches timed"
        dt = dt_eager
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    docs_per_s = world * batch * steps / dt
    flops = encoder_flops(batch * seq_len, seq_len, cfg)
    tf_per_gpu = flops * steps / dt / 1e12
    norms = out.norm(dim=1)
    return {
        "value": round(docs_per_s, 1),
        "unit": "docs/s",
        "ms_per_step": round(dt / steps * 1e3, 4),
        "launch_mode": "hip graph replay (one graph per forward)" if graph_note is None else "eager",
        "ms_per_step_eager": round(dt_eager / steps * 1e3, 4),
        "graph_note": graph_note,
        "dtype": "bf16",
        "config": {"workload": f"e5-small-v2-shaped encoder, batch {batch} x seq_len {seq_len} per GPU, "
                               "synthetic ids, random-init weights", "layers": cfg.num_hidden_layers},
        "roofline": {
            "bound": "mfma",
            "achieved": round(tf_per_gpu, 1),
            "peak": MFMA_BF16_PEAK_TF,
            "unit": "TFLOP/s",
            "frac": round(tf_per_gpu / MFMA_BF16_PEAK_TF, 4),
            "algorithmic_flops_per_step": flops,
        },
        "unit_norm_ok": bool(torch.allclose(norms, torch.ones_like(norms), atol=1e-3)),
        # rank 0's own rate on MS MARCO-shaped ragged lengths (per GPU, not aggregated)
        "ragged": bench_encode_ragged(enc, device) if ragged else None,
        # and end to end from Python strings (tokenise + H2D + forward + D2H) through StudentModel
        "text": bench_encode_text(enc, device) if text else None,
        # cfg 3's per-rank build: text -> HBM index shard, embeddings never leave the device
        "index_build": bench_index_build(enc, device) if text else None,
    }


def marco_like_lengths(n: int, seed: int = 7, max_len: int = 256) -> np.ndarray:
    """Token lengths of an MS MARCO-shaped passage set (clipped log-normal, mean ~75,
    median ~68, max 256 - an assumption of the survey, not pinned by the reference)."""
    rng = np.random.default_rng(seed)
    lens = rng.lognormal(mean=np.log(68.0), sigma=0.45, size=n)
    return np.clip(np.rint(lens), 8, max_len).astype(np.int64)


def bench_encode_ragged(enc: Mi355xSentenceEncoder, device, passes: int = 2, n_docs: int = 32768):
    """docs/s on ragged lengths through the varlen path (``encode_ragged``): whole sequences packed into
    256-token rows, launches cut by a token budget; inputs are host token ids (flat int32 stream +
    lengths), so the H2D copies of the ids are inside the timed region.  FLOPs are counted on real
    tokens only (attention term with each sequence's own length)."""
    cfg = enc.config
    lens = marco_like_lengths(n_docs).astype(np.int32)  # arrival order: NOT sorted
    rng = np.random.default_rng(11)
    flat = rng.integers(999, cfg.vocab_size, size=int(lens.sum()), dtype=np.int64).astype(np.int32)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    flat[starts] = 101
    out = torch.empty((n_docs, cfg.hidden_size), dtype=torch.float32, device=device)
    flops = float(sum(encoder_flops(int(l), int(l), cfg) for l in lens))
    enc.encode_ragged(flat, lens, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        enc.encode_ragged(flat, lens, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / passes
    return {
        "value": round(n_docs / dt, 1),
        "unit": "docs/s",
        "workload": f"{n_docs} passages in arrival order, lengths clipped log-normal (mean {lens.mean():.0f}, median "
                    f"{int(np.median(lens))}, max {int(lens.max())} tokens; SURVEY.md §8d assumption), packed into "
                    f"256-token rows (block-diagonal attention), host token ids in",
        "real_tokens_per_s": round(float(lens.sum()) / dt, 1),
        "padding_overhead": round(enc.last_encode_stats["padding_overhead"], 4),
        "mfma_frac_real_tokens": round(flops / dt / 1e12 / MFMA_BF16_PEAK_TF, 4),
    }


def synthetic_vocab(size: int = 30522):
    """A WordPiece vocabulary of the e5 size made of pronounceable pseudo-words (no real vocab is
    on disk offline): BERT's special-token ids, then whole words and ``##`` continuations."""
    rng = np.random.default_rng(5)
    cons, vow = "bcdfghjklmnprstvwz", "aeiou"
    words = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    words += [f"[unused{i}]" for i in range(99, 994)]   # ids 104 .. 998 as in bert-base-uncased
    words += list("abcdefghijklmnopqrstuvwxyz0123456789.,;:!?'-")
    seen = set(words)
    while len(words) < size:
        n_syl = int(rng.integers(1, 4))
        w = "".join(cons[rng.integers(len(cons))] + vow[rng.integers(len(vow))] for _ in range(n_syl))
        if rng.random() < 0.3:
            w = "##" + w
        if w not in seen:
            seen.add(w)
            words.append(w)
    return words


def synthetic_passages(vocab, n_docs: int, seed: int = 3):
    """MS MARCO-shaped English-like text over the synthetic vocabulary: word counts such that the
    WordPiece lengths follow ``marco_like_lengths`` (plus the ``passage: `` prefix of the e5 path)."""
    rng = np.random.default_rng(seed)
    whole = np.array([w for w in vocab[1000:] if not w.startswith("##") and len(w) > 1])
    lens = np.maximum(marco_like_lengths(n_docs, seed=seed) - 5, 1)
    picks = rng.integers(0, whole.size, size=int(lens.sum()))
    docs, pos = [], 0
    for n in lens:
        docs.append(" ".join(whole[picks[pos : pos + n]]))
        pos += int(n)
    return docs


def bench_encode_text(enc: Mi355xSentenceEncoder, device, n_docs: int = 32768, passes: int = 3):
    """docs/s of ``StudentModel.encode_documents(list_of_str)`` end to end - tokenise (background
    thread) + H2D + packed forward + D2H to NumPy - the call ``build_from_parquet`` makes
    (reference: scripts/build_faiss_index.py:55-62 with its default batch_size=32)."""
    from .encoder import build_wordpiece_tokenizer
    from .student import StudentModel

    vocab = synthetic_vocab(enc.config.vocab_size)
    enc.tokenizer = build_wordpiece_tokenizer(vocab)
    student = StudentModel.from_encoder(enc, "e5-small-v2-synthetic")
    docs = synthetic_passages(vocab, n_docs)
    emb = student.encode_documents(docs, batch_size=32)   # untimed warm-up pass: staging buffers, thread pools
    assert emb.shape == (n_docs, enc.config.hidden_size)
    t0 = time.perf_counter()
    for _ in range(passes):
        emb = student.encode_documents(docs, batch_size=32)
    dt = (time.perf_counter() - t0) / passes
    stats = enc.last_encode_stats
    t1 = time.perf_counter()
    enc._tokenize_flat(["passage: " + d for d in docs[:8192]])
    tok_rate = 8192 / (time.perf_counter() - t1)
    return {
        "value": round(n_docs / dt, 1),
        "unit": "docs/s",
        "workload": f"StudentModel.encode_documents over {n_docs} synthetic-vocabulary passages (Python str in, "
                    f"NumPy out, batch_size=32 as the reference CLI passes it)",
        "mean_tokens": round(stats["real_tokens"] / n_docs, 1),
        "padding_overhead": round(stats["padding_overhead"], 4),
        "tokenizer_alone_docs_per_s": round(tok_rate, 1),
        "unit_norm_ok": bool(np.allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-3)),
    }


def bench_index_build(enc: "Mi355xSentenceEncoder", device, n_docs: int = 131072):
    """BASELINE cfg 3's build step as ONE rank runs it (scripts/build_faiss_index.py:45-72 per shard, sharded_index.
    build_sharded): MS MARCO-shaped passages (Python str) -> tokenise -> packed encoder -> embeddings stay in HBM ->
    ``FAISSIndexBuilder.add`` (normalise + tile) -> screening sidecar.  No host round trip of the embeddings.  A rank of
    the 8-GPU job owns 1 105 228 passages: the projected shard build time is reported beside the measured rate."""
    from .encoder import build_wordpiece_tokenizer
    from .index import FAISSIndexBuilder
    from .student import StudentModel

    vocab = synthetic_vocab(enc.config.vocab_size)
    if enc.tokenizer is None:
        enc.tokenizer = build_wordpiece_tokenizer(vocab)
    student = StudentModel.from_encoder(enc, "e5-small-v2-synthetic")
    docs = synthetic_passages(vocab, n_docs, seed=31)
    student.encode_documents_device(docs[:8192])      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index = FAISSIndexBuilder(embedding_dim=enc.config.hidden_size, index_type="HNSW", metric="cosine", device=str(device))
    index.reserve(n_docs)
    slab = 65536
    for lo in range(0, n_docs, slab):
        index.add(student.encode_documents_device(docs[lo : lo + slab]))
    q = torch.nn.functional.normalize(torch.randn((64, enc.config.hidden_size), device=device), dim=1)
    index.search_device(q, 10)                          # builds the bf16 screening sidecar
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {
        "value": round(n_docs / dt, 1), "unit": "docs/s",
        "workload": f"{n_docs} synthetic-vocabulary passages (str) -> embeddings -> HBM index shard + screening sidecar, on one GPU",
        "seconds": round(dt, 3),
        "projected_seconds_for_a_1105228_passage_shard": round(1105228 / (n_docs / dt), 1),
        "ntotal": int(index.ntotal),
    }


def bench_kd_step(device, tuples: int = 32, docs_per_query: int = 8, q_len: int = 32, d_len: int = 256,
                  steps: int = 5, warmup: int = 2):
    """BASELINE cfg 4: one KD training step = student forward over ``tuples`` queries and their
    (positive + 7 hard negatives) passages with saved activations, scores ``q . d``, the fused
    Margin-MSE + listwise-KL + InfoNCE loss, backward through the encoder, AdamW update (reference
    step: src/kd/train.py:176-210, batched over the tuples instead of one query at a time).
    e5-small-v2 architecture, random-init weights, synthetic ids, bf16 compute / fp32 gradients.
    FLOPs: forward FLOPs of SURVEY.md section 8(d) on the step's tokens x 3 (backward = 2 x forward)."""
    from .losses import CombinedKDLoss
    from .training import TrainableEncoder
    from .weights import synthetic_state_dict

    cfg = BertConfig()
    model = TrainableEncoder(cfg, synthetic_state_dict(cfg), device)
    # torch.optim.AdamW as the reference uses it (src/kd/train.py:131-135), in its fused form: one pass over the 33 M
    # parameters instead of ~10 foreach passes (~1.5 ms of a 30 ms step)
    opt_kw = {"fused": True}
    try:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-5, **opt_kw)
    except (RuntimeError, TypeError, ValueError):
        opt_kw = {}
        opt = torch.optim.AdamW(model.parameters(), lr=1e-5)
    loss_fn = CombinedKDLoss()
    q_ids, q_mask = synthetic_ids(tuples, q_len, cfg.vocab_size, device, seed=1)
    d_ids, d_mask = synthetic_ids(tuples * docs_per_query, d_len, cfg.vocab_size, device, seed=2)
    g = torch.Generator(device=device).manual_seed(3)
    teacher = torch.randn((tuples, docs_per_query), generator=g, device=device) * 3.0

    def step():
        opt.zero_grad(set_to_none=True)
        q = model(q_ids, q_mask)
        d = model(d_ids, d_mask).view(tuples, docs_per_query, -1)
        scores = torch.einsum("th,tdh->td", q, d)
        out = loss_fn(scores, teacher)
        out["loss"].backward()
        opt.step()
        return out["loss"]

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt_eager = (time.perf_counter() - t0) / steps
    # the same step captured into ONE HIP graph and replayed (training.GraphedStep): every launch of the step - both
    # encoder forwards, the loss, the backward pass, AdamW - is inside the graph and inside the timed region
    graph_note = None
    try:
        from .training import GraphedStep

        opt = torch.optim.AdamW(model.parameters(), lr=1e-5, capturable=True, **opt_kw)
        graphed = GraphedStep(step, warmup=2)
        graphed()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = graphed()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    except Exception as exc:  # noqa: BLE001 - reported in the line; the eager time stands
        graph_note = f"graph capture failed ({type(exc).__name__}: {exc}); eager launches timed"
        dt = dt_eager
    flops = 3.0 * (encoder_flops(tuples * q_len, q_len, cfg) + encoder_flops(tuples * docs_per_query * d_len, d_len, cfg))
    return {
        "value": round(tuples / dt, 1),
        "unit": "tuples/s",
        "ms_per_step": round(dt * 1e3, 3),
        "launch_mode": "hip graph replay (one graph per step)" if graph_note is None else "eager",
        "optimizer": "torch.optim.AdamW(fused=True)" if opt_kw else "torch.optim.AdamW",
        "ms_per_step_eager": round(dt_eager * 1e3, 3),
        "graph_note": graph_note,
        "dtype": "bf16",
        "workload": f"{tuples} (query, positive, {docs_per_query - 1} hard-negative) tuples per step: queries {q_len} tokens, "
                    f"passages {d_len} tokens; forward + fused KD loss + backward + AdamW",
        "tokens_per_s": round((tuples * q_len + tuples * docs_per_query * d_len) / dt, 1),
        "final_loss": float(loss.detach()),
        "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                     "frac": round(flops / dt / 1e12 / MFMA_BF16_PEAK_TF, 4), "algorithmic_flops_per_step": flops},
    }


def bench_teacher(device, world: int, steps: int, warmup: int, barrier, pairs: int = 128, seq_len: int = 256):
    """BASELINE cfg 5 model: XLM-R-large-shaped cross-encoder (24 layers, hidden 1024, 16 heads, FFN 4096;
    bge-reranker-large) scoring (query, passage) pairs of ``seq_len`` tokens, ``pairs`` per rank per step;
    pure data parallel (every rank scores its own pairs, no communication).  Random-init weights drawn on
    the device, synthetic ids.  FLOPs per token: 24 (2 (4 H^2 + 2 H F) + 4 S H)."""
    import torch.distributed as dist

    from .teacher import TeacherConfig, TeacherModel

    cfg = TeacherConfig()
    teacher = TeacherModel.from_random_device(cfg, str(device), seed=int(device.index or 0))
    g = torch.Generator(device=device).manual_seed(5)
    ids = torch.randint(4, cfg.vocab_size, (pairs, seq_len), generator=g, device=device, dtype=torch.int32)
    ids[:, 0] = 0
    ids[:, -1] = 2
    mask = torch.ones_like(ids)
    out = torch.empty(pairs, dtype=torch.float32, device=device)
    for _ in range(max(warmup, 1)):
        teacher.score_token_ids(ids, mask, out=out)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        teacher.score_token_ids(ids, mask, out=out)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    h, f = cfg.hidden_size, cfg.intermediate_size
    flops = pairs * seq_len * cfg.num_hidden_layers * (2.0 * (4 * h * h + 2 * h * f) + 4.0 * seq_len * h)
    tf = flops * steps / dt / 1e12
    # the same step with every product on this repo's own MFMA kernels (sskd_gemm_backend(1)): what the hand-written
    # path alone delivers, beside the default in which the three plain products of a layer go through hipBLASLt
    own = None
    if world == 1:
        from . import _native

        lib = _native.load()
        try:
            lib.sskd_gemm_backend(1)
            teacher.score_token_ids(ids, mask, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(2, steps // 2)):
                teacher.score_token_ids(ids, mask, out=out)
            torch.cuda.synchronize()
            dt_own = (time.perf_counter() - t0) / max(2, steps // 2)
        finally:
            lib.sskd_gemm_backend(0)
    text = bench_teacher_text(teacher) if (world == 1 or dist.get_rank() == 0) else None
    h, f = cfg.hidden_size, cfg.intermediate_size
    if world == 1:
        flops1 = pairs * seq_len * cfg.num_hidden_layers * (2.0 * (4 * h * h + 2 * h * f) + 4.0 * seq_len * h)
        own = {"ms_per_step": round(dt_own * 1e3, 3), "mfma_bf16_frac": round(flops1 / dt_own / 1e12 / MFMA_BF16_PEAK_TF, 4)}
    return {
        "value": round(world * pairs * steps / dt, 1),
        "unit": "pairs/s",
        "ms_per_step": round(dt / steps * 1e3, 3),
        "gemm_backend": "QKV / attention output / FFN2: hipBLASLt (plain products); FFN1 + erf-GELU, attention, LayerNorm, head: this repo's kernels",
        "own_kernels_only": own,
        "dtype": "bf16",
        "workload": f"XLM-R-large-shaped cross-encoder, {pairs} pairs x {seq_len} tokens per GPU per step, random-init weights",
        "finite": bool(torch.isfinite(out).all()),
        "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                     "frac": round(tf / MFMA_BF16_PEAK_TF, 4), "algorithmic_flops_per_step": flops},
        "text": text,
    }


def bench_teacher_text(teacher, n_pairs: int = 8192):
    """``TeacherModel.score(list of (query, passage) strings)`` end to end - what TeacherMiner and the /search rerank
    branch call (reference: src/mining/miners.py:135-137, src/serve/app.py:325-326) - on MS MARCO-shaped pairs over a
    synthetic vocabulary, next to the same launches fed from cached token ids (the GPU-only rate) and the tokenizer
    alone: the text path should sit within ~15 % of the slower of the two."""
    from .encoder import build_wordpiece_tokenizer

    vocab = synthetic_vocab(30522)
    teacher.tokenizer = build_wordpiece_tokenizer(vocab)
    docs = synthetic_passages(vocab, n_pairs, seed=21)
    rng = np.random.default_rng(22)
    whole = [w for w in vocab[1000:4000] if not w.startswith("##") and len(w) > 1]
    pairs = [(" ".join(rng.choice(whole, size=int(rng.integers(4, 12)))), d) for d in docs]
    teacher.score(pairs[:512])                                   # warm-up: workspace, tokenizer threads
    t0 = time.perf_counter()
    scores = teacher.score(pairs)
    t_text = time.perf_counter() - t0
    t0 = time.perf_counter()
    cached = [teacher.tokenize_pairs(pairs[c : c + teacher.SCORE_CHUNK]) for c in range(0, n_pairs, teacher.SCORE_CHUNK)]
    t_tok = time.perf_counter() - t0
    real_tokenize, it = teacher.tokenize_pairs, iter(cached)
    teacher.tokenize_pairs = lambda chunk: next(it)
    try:
        t0 = time.perf_counter()
        again = teacher.score(pairs)
        t_gpu = time.perf_counter() - t0
    finally:
        teacher.tokenize_pairs = real_tokenize
    mean_tokens = float(np.mean([m.sum() for _, m in cached])) * len(cached) / n_pairs
    return {
        "value": round(n_pairs / t_text, 1), "unit": "pairs/s",
        "workload": f"TeacherModel.score over {n_pairs} (query, passage) string pairs, mean {mean_tokens:.0f} tokens per pair "
                    f"(synthetic vocabulary), tokenise + H2D + forward + D2H",
        "from_cached_token_ids_pairs_per_s": round(n_pairs / t_gpu, 1),
        "tokenizer_alone_pairs_per_s": round(n_pairs / t_tok, 1),
        "same_scores_both_ways": bool(np.allclose(scores, again, atol=1e-3)),
    }
