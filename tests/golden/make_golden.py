#!/usr/bin/env python3
"""Generates the golden fixtures in this directory (run in the build container, CPU only).

The reference keeps no golden vectors for the hot path, and its own hot-path
source files / engines are absent here, so fixtures come from:

* search  — inputs built exactly like the reference's only index fixture (tests/conftest.py:65-73:
            ``np.random.seed(42); randn(10, 384)`` L2-normalised rows in an ``IndexFlatIP``) and the
            BASELINE cfg-1 shape (1 000 x 100, k = 10); expected outputs from the reference's exact
            search idiom (``np.matmul`` + ``argsort``: src/kd/eval.py:86, scripts/simple_eval.py:25,35)
            as restated in oracle/search.py, in BLAS order and in the kernel's fma order.
* KD loss — the reference's own ``src/kd/losses.py`` (importable here; see ``make_kd_loss``): losses and
            autograd gradients for the §8(f) loss row, i.e. a fixture produced by reference code.
* encoder — ``transformers.BertModel`` (the module sentence-transformers executes for the reference's
            StudentModel) built from an in-memory ``BertConfig`` — nothing is downloaded — loaded
            with the deterministic synthetic weights of semantic-search-kd_amd/weights.py.  The
            oracle restatement (oracle/encoder.py) is asserted equal to it here, and the
            embeddings are stored.  Weights are NOT stored: tests regenerate them from the recipe.

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))

from oracle import encoder as enc_oracle  # noqa: E402
from oracle import search as oracle  # noqa: E402
from semantic_search_kd_amd.weights import BertConfig, synthetic_state_dict  # noqa: E402


def make_search_small():
    np.random.seed(42)  # tests/conftest.py:69-72
    emb = np.random.randn(10, 384).astype(np.float32)
    emb = emb / np.linalg.norm(emb, axis=1, keepdims=True)
    q = oracle.seeded_unit_rows(5, 384, 99)
    q[0] = emb[3]
    out = {"corpus": emb, "queries": q}
    for k in (1, 3, 10, 20):
        bs, bi = oracle.topk_blas(q, emb, k)
        fs, fi = oracle.topk_fma(q, emb, k)
        assert np.array_equal(bi, fi) and np.abs(bs - fs)[bi >= 0].max() < 1e-6
        out[f"blas_scores_k{k}"], out[f"ids_k{k}"], out[f"fma_scores_k{k}"] = bs, bi, fs
    np.savez_compressed(HERE / "search_small.npz", **out)


def make_search_1k():
    c = oracle.seeded_unit_rows(1000, 384, 1234)
    q = oracle.seeded_unit_rows(100, 384, 4321)
    bs, bi = oracle.topk_blas(q, c, 10)
    fs, fi = oracle.topk_fma(q, c, 10)
    assert np.array_equal(bi, fi)
    # inputs are regenerated from the seeds by the tests; only expected outputs are stored
    np.savez_compressed(HERE / "search_1k.npz", blas_scores=bs, ids=bi, fma_scores=fs,
                        corpus_checksum=np.float64(c.astype(np.float64).sum()),
                        queries_checksum=np.float64(q.astype(np.float64).sum()))


def _register_reference_stubs():
    """Make the reference's evaluation modules importable in this container: ``loguru`` (logging
    only) is absent and ``src/models/`` is missing from the checkout - the two
    model classes are type annotations in the modules driven below, never called."""
    import types

    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    stub = types.ModuleType("loguru")

    class _Silent:
        def __getattr__(self, name):
            return lambda *a, **k: None

    stub.logger = _Silent()
    sys.modules.setdefault("loguru", stub)
    models = types.ModuleType("src.models")
    models.__path__ = []
    student = types.ModuleType("src.models.student")
    student.StudentModel = type("StudentModel", (), {})
    teacher = types.ModuleType("src.models.teacher")
    teacher.TeacherModel = type("TeacherModel", (), {})
    sys.modules.setdefault("src.models", models)
    sys.modules.setdefault("src.models.student", student)
    sys.modules.setdefault("src.models.teacher", teacher)


class _RecordingArray(np.ndarray):
    """ndarray that records what ``np.matmul`` returns when the REFERENCE multiplies it
    (scripts/simple_eval.py:25): the similarity matrix its argsort then ranks."""

    matmul_results: list = []

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        plain = [np.asarray(i) for i in inputs]
        out = getattr(ufunc, method)(*plain, **kwargs)
        if ufunc is np.matmul:
            _RecordingArray.matmul_results.append(np.array(out, copy=True))
        return out


class _SpyLabels:
    """A relevance-label row that records which document indices the reference looks up, in order
    (``labels[idx] if idx < len(labels) else 0`` for idx in ``np.argsort(sims)[::-1][:k]`` -
    scripts/simple_eval.py:35-36, src/kd/eval.py:86-87): exactly the reference's top-k ids."""

    def __init__(self, n, relevant):
        self.n, self.relevant, self.calls = n, relevant, []

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        self.calls.append(int(idx))
        return 1 if int(idx) == self.relevant else 0


class _FixedEmbeddingModel:
    """Duck-typed stand-in for StudentModel (the class is absent from the checkout): returns the
    committed seed embeddings for "q<i>" / "d<i>" strings.  ``compute_similarity`` is the q @ d.T
    every call site assumes (tests/test_student_model.py:116-124)."""

    def __init__(self, q, c):
        self.q, self.c = q, c

    def encode_queries(self, queries, **kw):
        return self.q[[int(s[1:]) for s in queries]].view(_RecordingArray)

    def encode_documents(self, docs, **kw):
        return self.c[[int(s[1:]) for s in docs]].view(_RecordingArray)

    def compute_similarity(self, q, d):
        return np.matmul(np.asarray(q), np.asarray(d).T)


def _reference_topk(q, c, k_values, relevant):
    """Top-k ids and the similarity matrix as computed by the REFERENCE'S OWN CODE:
    scripts/simple_eval.py::evaluate_model (np.matmul + np.argsort(...)[::-1][:k]) and
    src/kd/eval.py::KDEvaluator.evaluate_retrieval (compute_similarity + the same argsort)."""
    import importlib.util

    _register_reference_stubs()
    spec = importlib.util.spec_from_file_location("ref_simple_eval", "/root/reference/scripts/simple_eval.py")
    simple_eval = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(simple_eval)
    from src.kd.eval import KDEvaluator

    nq, n = q.shape[0], c.shape[0]
    model = _FixedEmbeddingModel(q, c)
    queries, corpus = [f"q{i}" for i in range(nq)], [f"d{i}" for i in range(n)]
    out = {}
    for tag, run in (
        ("simple_eval", lambda labels: simple_eval.evaluate_model(model, queries, corpus, labels, k_values=list(k_values))),
        ("kd_eval", lambda labels: KDEvaluator(model).evaluate_retrieval(queries, corpus, labels, k_values=list(k_values))),
    ):
        labels = [_SpyLabels(n, int(relevant[i])) for i in range(nq)]
        _RecordingArray.matmul_results.clear()
        metrics = run(labels)
        ids = {}
        for i, lab in enumerate(labels):
            pos = 0
            for k in k_values:  # the reference walks k_values in order, every query per k
                kk = min(k, n)
                ids.setdefault(k, []).append(lab.calls[pos : pos + kk])
                pos += kk
            assert pos == len(lab.calls)
        out[tag] = ({k: np.array(v, np.int64) for k, v in ids.items()}, {k: float(v) for k, v in metrics.items()})
        if tag == "simple_eval":
            assert len(_RecordingArray.matmul_results) == 1
            out["similarities"] = _RecordingArray.matmul_results[0].astype(np.float32)
    a, b = out["simple_eval"][0], out["kd_eval"][0]
    assert all(np.array_equal(a[k], b[k]) for k in k_values), "the two reference call sites disagree"
    return out


def make_search_ref():
    """Search fixtures produced by REFERENCE-HELD code (this pins oracle/search.py): the ids are
    what the reference's own argsort idiom selected, the scores what its own np.matmul produced.
    mrr@k of the reference's metrics is cross-checked: with one relevant document per query it
    equals 1 / rank of that document inside the recorded top-k (src/utils/metrics.py:40-55)."""
    cases = {}
    np.random.seed(42)  # tests/conftest.py:69-72
    emb = np.random.randn(10, 384).astype(np.float32)
    emb = emb / np.linalg.norm(emb, axis=1, keepdims=True)
    q = oracle.seeded_unit_rows(5, 384, 99)
    q[0] = emb[3]
    cases["small"] = (q, emb, (1, 3, 10, 20))
    cases["1k"] = (oracle.seeded_unit_rows(100, 384, 4321), oracle.seeded_unit_rows(1000, 384, 1234), (1, 5, 10))
    for tag, (q, c, ks) in cases.items():
        nq, n = q.shape[0], c.shape[0]
        relevant = (np.arange(nq) * 7 + 3) % n  # an arbitrary planted "relevant" document per query
        res = _reference_topk(q, c, ks, relevant)
        ids, metrics = res["simple_eval"]
        sims = res["similarities"]
        store = {"similarities_checksum": np.float64(sims.astype(np.float64).sum()), "relevant": relevant}
        for k in ks:
            ref_ids = ids[k]                      # [nq, min(k, n)]
            ref_scores = np.take_along_axis(sims, ref_ids, axis=1)
            assert (np.diff(ref_scores, axis=1) <= 0).all()
            # the reference's mrr@k is 1 / rank of the planted document
            rr = [(1.0 / (list(r).index(rel) + 1)) if rel in r else 0.0 for r, rel in zip(ref_ids, relevant)]
            assert abs(np.mean(rr) - metrics[f"mrr@{k}"]) < 1e-12, (tag, k)
            # the restatement agrees with the reference on ids (no exact ties in these inputs) and scores
            os_, oi = oracle.topk_blas(q, c, k)
            kk = ref_ids.shape[1]
            assert np.array_equal(oi[:, :kk], ref_ids) and (oi[:, kk:] == -1).all(), (tag, k)
            assert np.abs(os_[:, :kk] - ref_scores).max() <= 1e-6
            fs, fi = oracle.topk_fma(q, c, k)
            assert np.array_equal(fi[:, :kk], ref_ids) and np.abs(fs[:, :kk] - ref_scores).max() <= 1e-6
            store[f"ref_ids_k{k}"], store[f"ref_scores_k{k}"] = ref_ids, ref_scores
            store[f"ref_mrr_k{k}"] = np.float64(metrics[f"mrr@{k}"])
            store[f"ref_ndcg_k{k}"] = np.float64(metrics[f"ndcg@{k}"])
        gaps = -np.diff(np.sort(sims, axis=1)[:, ::-1][:, : min(max(ks) + 1, n)], axis=1)
        store["min_rank_gap"] = np.float64(gaps.min())
        # fp32 summation orders (BLAS here, fma chain on the GPU) differ by ~1e-7 on unit vectors:
        # queries with two of their top (k+1) scores closer than 2e-6 are listed; everywhere else
        # id parity is unconditional
        store["near_tie_queries"] = np.where((gaps < 2e-6).any(axis=1))[0].astype(np.int64)
        if tag == "small":
            store["corpus"], store["queries"] = c, q
        print(f"[search_ref_{tag}] reference code produced ids for k={ks}; min rank gap {gaps.min():.2e}, near-tie queries {store['near_tie_queries'].tolist()}")
        np.savez_compressed(HERE / f"search_ref_{tag}.npz", **store)


class HashedEmbeddingStudent:
    """Deterministic stand-in student for miner fixtures: the embedding of a text is a unit vector
    seeded by crc32(text) (+ the e5 role prefix the real StudentModel would prepend); similarity in
    float64 so that batched and per-query products agree to the bit."""

    dim = 64

    def _emb(self, texts):
        import zlib

        out = np.empty((len(texts), self.dim), np.float32)
        for i, t in enumerate(texts):
            g = np.random.Generator(np.random.PCG64(zlib.crc32(t.encode("utf-8"))))
            v = g.standard_normal(self.dim)
            out[i] = (v / np.linalg.norm(v)).astype(np.float32)
        return out

    def encode_queries(self, queries, **kw):
        return self._emb(["query: " + q for q in ([queries] if isinstance(queries, str) else queries)])

    def encode_documents(self, docs, **kw):
        return self._emb(["passage: " + d for d in ([docs] if isinstance(docs, str) else docs)])

    def compute_similarity(self, q, d):
        return (np.asarray(q, np.float64) @ np.asarray(d, np.float64).T).astype(np.float32)


def ance_case():
    """Inputs of the ANCE fixture: topical word pools so that some candidates land inside the margin."""
    rng = np.random.default_rng(12)
    words = [f"w{i}" for i in range(40)]
    docs = {f"d{i}": " ".join(rng.choice(words, size=int(rng.integers(3, 9)))) for i in range(60)}
    docs["d7"] = ""  # empty text
    queries = [" ".join(rng.choice(words, size=4)) for _ in range(12)]
    positives = [[f"d{int(j)}" for j in rng.choice(60, size=int(rng.integers(0, 3)), replace=False)] for _ in queries]
    candidates = [[f"d{int(j)}" for j in rng.choice(60, size=int(rng.integers(0, 25)), replace=False)] for _ in queries]
    candidates[3] = candidates[3] + ["missing-id"]  # not in the text table -> "" (miners.py:219)
    return queries, positives, candidates, docs


def make_ance():
    """Hard negatives chosen by the REFERENCE'S OWN ``ANCEMiner.mine`` (src/mining/miners.py:184-253)
    for a deterministic stand-in student.  ``rank_bm25`` (used by the sibling BM25 classes of that
    module, never by ANCEMiner) is absent here: an empty placeholder module lets the import through."""
    import json
    import types

    _register_reference_stubs()
    if "rank_bm25" not in sys.modules:
        ph = types.ModuleType("rank_bm25")
        ph.BM25Okapi = type("BM25Okapi", (), {})
        sys.modules["rank_bm25"] = ph
    from src.mining.miners import ANCEMiner as RefMiner

    queries, positives, candidates, docs = ance_case()
    out = {}
    for margin in (0.1, 0.3, 0.0):
        for top_k in (5, 2):
            got = RefMiner(HashedEmbeddingStudent(), margin=margin).mine(queries, positives, candidates, docs, docs, top_k=top_k)
            out[f"margin{margin}_k{top_k}"] = got
    (HERE / "ance_mining.json").write_text(json.dumps(out, indent=0, sort_keys=True) + "\n")
    print("[ance] reference ANCEMiner.mine:", {k: sum(len(x) for x in v) for k, v in out.items()}, "negatives")


class HashedTeacher:
    """Deterministic stand-in teacher for the mining fixture: a score that depends only on the (query, text) strings
    (crc32 -> [-4, 4)), with duplicates of one candidate text giving exact score ties; ``get_confidence`` = sigmoid,
    as the product's TeacherModel.  Records how ``score`` was called."""

    def __init__(self):
        self.calls = []

    def score(self, pairs, batch_size=32):
        import zlib

        self.calls.append((len(pairs), batch_size))
        return [((zlib.crc32((q + "\x00" + t).encode()) % 8000) / 1000.0) - 4.0 for q, t in pairs]

    @staticmethod
    def get_confidence(score):
        import math

        return 1.0 / (1.0 + math.exp(-float(score)))


def teacher_mining_case():
    queries, _, candidates, docs = ance_case()
    docs = dict(docs)
    docs["d11"] = docs["d12"]              # identical texts -> exact score ties: the stable sort decides
    candidates = [list(c) for c in candidates]
    candidates[1] = candidates[1] + ["d11", "d12"]
    return queries, candidates, docs


def make_teacher_mining():
    """Hard negatives and scores chosen by the REFERENCE'S OWN ``TeacherMiner.mine`` (src/mining/miners.py:104-158)
    for a deterministic stand-in teacher."""
    import json
    import types

    _register_reference_stubs()
    if "rank_bm25" not in sys.modules:
        ph = types.ModuleType("rank_bm25")
        ph.BM25Okapi = type("BM25Okapi", (), {})
        sys.modules["rank_bm25"] = ph
    from src.mining.miners import TeacherMiner as RefMiner

    queries, candidates, docs = teacher_mining_case()
    out = {}
    for thr in (0.6, 0.5, 0.9):
        for top_k in (10, 3):
            ids, scores = RefMiner(HashedTeacher(), confidence_threshold=thr).mine(queries, candidates, docs, top_k=top_k)
            out[f"thr{thr}_k{top_k}"] = {"ids": ids, "scores": scores}
    (HERE / "teacher_mining.json").write_text(json.dumps(out, indent=0, sort_keys=True) + "\n")
    print("[teacher_mining] reference TeacherMiner.mine:", {k: sum(len(x) for x in v["ids"]) for k, v in out.items()}, "negatives")


def make_pool_norm():
    g = np.random.Generator(np.random.PCG64(7))
    h = g.standard_normal((8, 64, 384), dtype=np.float32)
    lens = [64, 50, 33, 20, 12, 7, 1, 64]
    mask = np.zeros((8, 64), np.int32)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    e = enc_oracle.mean_pool_normalize(h, mask, True)
    e_raw = enc_oracle.mean_pool_normalize(h, mask, False)
    # torch restatement of sentence_transformers.models.Pooling(mean) + Normalize
    ht, mt = torch.from_numpy(h), torch.from_numpy(mask).float().unsqueeze(-1)
    ref = (ht * mt).sum(1) / torch.clamp(mt.sum(1), min=1e-9)
    assert np.abs(ref.numpy() - e_raw).max() < 1e-6
    assert np.abs(torch.nn.functional.normalize(ref, p=2, dim=1).numpy() - e).max() < 1e-6
    assert np.abs(oracle.pool_normalize(h, mask, True) - e).max() < 1e-6
    np.savez_compressed(HERE / "pool_norm.npz", seed=7, lengths=np.array(lens), pooled=e_raw, normalized=e)


def hf_bert(cfg: BertConfig, sd):
    from transformers import BertConfig as HFConfig
    from transformers import BertModel

    hf_cfg = HFConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
        max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
        layer_norm_eps=cfg.layer_norm_eps, hidden_act="gelu", hidden_dropout_prob=0.0,
        attention_probs_dropout_prob=0.0,
    )
    model = BertModel(hf_cfg, add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m or "pooler" in m for m in missing), missing
    return model


def make_bert(layers: int, tag: str, stress: bool = False):
    """``stress``: trained-checkpoint-like hard cases (weights.synthetic_state_dict(stress=True)):
    peaky attention rows, LayerNorm gains in [0.3, 3], massive-activation channels, and sequences
    long enough (up to 200 tokens = 7 key tiles) for the online-softmax rescale to fire."""
    cfg = BertConfig(num_hidden_layers=layers)
    sd = synthetic_state_dict(cfg, stress=stress)
    lengths = [200, 131, 64, 5] if stress else [48, 31, 17, 5]
    ids, mask = enc_oracle.synthetic_token_ids(4, max(lengths), seed=11 + layers, lengths=lengths)
    model = hf_bert(cfg, sd)
    with torch.no_grad():
        out = model(
            input_ids=torch.from_numpy(ids).long(), attention_mask=torch.from_numpy(mask).long(),
            output_hidden_states=True,
        )
    hf_hidden = [h.numpy() for h in out.hidden_states]
    ours = enc_oracle.bert_hidden_states(sd, ids, mask, layers, return_all=True)
    m = mask.astype(bool)
    worst = max(float(np.abs(a[m] - b[m]).max()) for a, b in zip(hf_hidden, ours))
    print(f"[{tag}] oracle vs transformers.BertModel: max |diff| over real tokens = {worst:.3e}")
    assert worst < (2e-3 if stress else 2e-4), worst
    e_hf = enc_oracle.mean_pool_normalize(hf_hidden[-1], mask, True)
    e_or = enc_oracle.encode_token_ids(sd, ids, mask, layers)
    assert np.abs(e_hf - e_or).max() < 1e-5
    np.savez_compressed(
        HERE / f"bert_{tag}.npz",
        layers=layers, input_ids=ids, attention_mask=mask, embeddings=e_hf,
        layer_mean_abs=np.array([np.abs(h[m]).mean() for h in hf_hidden], np.float64),
        last_hidden_cls=hf_hidden[-1][:, 0, :].astype(np.float32),
        oracle_vs_hf_max_abs=np.float64(worst),
    )


GRAD_CFG = dict(vocab_size=600, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                max_position_embeddings=64)


def grad_case():
    cfg = BertConfig(**GRAD_CFG)
    sd = synthetic_state_dict(cfg)
    ids, mask = enc_oracle.synthetic_token_ids(5, 40, seed=31, vocab=cfg.vocab_size, lengths=[40, 33, 17, 8, 2])
    g = np.random.Generator(np.random.PCG64(32))
    probe = g.standard_normal((5, cfg.hidden_size)).astype(np.float32)  # loss = sum(embeddings * probe)
    return cfg, sd, ids, mask, probe


def make_bert_grads():
    """Parameter gradients of ``transformers.BertModel`` + mean-pool + L2-normalise by torch autograd
    (the reference's training path: src/kd/train.py:176-210 backpropagates through exactly these
    modules) on a small synthetic model; the oracle restatement's autograd is asserted equal here.
    Stored in float16 (the test gate is a cosine per parameter tensor)."""
    cfg, sd, ids, mask, probe = grad_case()
    model = hf_bert(cfg, sd).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    out = model(input_ids=torch.from_numpy(ids).long(), attention_mask=torch.from_numpy(mask).long())
    h = out.last_hidden_state
    m = torch.from_numpy(mask).float()[..., None]
    e = (h * m).sum(1) / torch.clamp(m.sum(1), min=1e-9)
    e = torch.nn.functional.normalize(e, p=2, dim=1)
    (e * torch.from_numpy(probe)).sum().backward()
    hf_grads = {k: p.grad.numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
    t = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd.items()}
    eo = enc_oracle.embeddings_torch(t, ids, mask, cfg.num_hidden_layers, cfg.num_attention_heads)
    assert np.abs(eo.detach().numpy() - e.detach().numpy()).max() < 1e-5
    (eo * torch.from_numpy(probe)).sum().backward()
    store = {"embeddings": e.detach().numpy()}
    for k, g in hf_grads.items():
        og = t[k].grad.numpy()
        denom = np.abs(g).max() + 1e-12
        assert np.abs(og - g).max() <= 2e-4 * denom + 1e-7, (k, np.abs(og - g).max(), denom)
        store["grad:" + k] = (g / denom).astype(np.float16)   # scaled to max |g| = 1 (fp16 range)
        store["amax:" + k] = np.float64(denom)
    np.savez_compressed(HERE / "bert_grads_small.npz", **store)
    print(f"[bert_grads_small] {len(hf_grads)} parameter tensors, oracle autograd == transformers autograd")


TEACHER_SMALL = dict(vocab_size=800, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=512,
                     max_position_embeddings=130)


def teacher_case():
    from semantic_search_kd_amd.teacher import TeacherConfig, synthetic_teacher_state_dict

    cfg = TeacherConfig(**TEACHER_SMALL)
    sd = synthetic_teacher_state_dict(cfg)
    lengths = [96, 70, 33, 12, 5]
    ids, mask = enc_oracle.synthetic_token_ids(5, 96, seed=51, vocab=cfg.vocab_size, lengths=lengths)
    ids = np.where(mask == 1, np.maximum(ids, 4), cfg.pad_token_id).astype(np.int32)  # XLM-R: <s>=0, <pad>=1, </s>=2
    ids[:, 0] = 0
    for b, n in enumerate(lengths):
        ids[b, n - 1] = 2
    return cfg, sd, ids, mask


def _xlmr_logits(cfg, sd, ids, mask):
    """``transformers.XLMRobertaForSequenceClassification`` (one label) from an IN-MEMORY config + the given weights"""
    from transformers import XLMRobertaConfig, XLMRobertaForSequenceClassification

    hf_cfg = XLMRobertaConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
        max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
        layer_norm_eps=cfg.layer_norm_eps, hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
        classifier_dropout=0.0, num_labels=1, pad_token_id=cfg.pad_token_id, bos_token_id=0, eos_token_id=2,
    )
    model = XLMRobertaForSequenceClassification(hf_cfg).eval()
    hf_sd = {("roberta." + k if not k.startswith("classifier.") else k): torch.from_numpy(v) for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(hf_sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m for m in missing), missing
    with torch.no_grad():
        return model(input_ids=torch.from_numpy(ids).long(), attention_mask=torch.from_numpy(mask).long()).logits[:, 0].numpy()


def make_xlmr():
    """Logits of ``transformers.XLMRobertaForSequenceClassification`` (num_labels = 1: the architecture
    of the reference's bge-reranker-large teacher) built from an in-memory config on synthetic weights;
    the oracle restatement (oracle/teacher.py) is asserted equal here."""
    from oracle import teacher as teacher_oracle

    cfg, sd, ids, mask = teacher_case()
    want = _xlmr_logits(cfg, sd, ids, mask)
    got = teacher_oracle.logits(sd, ids, mask, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.layer_norm_eps, cfg.pad_token_id)
    print(f"[xlmr_small] oracle vs transformers logits: max |diff| = {np.abs(got - want).max():.3e}; logits {want}")
    assert np.abs(got - want).max() < 1e-5
    np.savez_compressed(HERE / "xlmr_small.npz", input_ids=ids, attention_mask=mask, logits=want)


TEACHER_SPREAD = dict(vocab_size=800, hidden_size=128, num_hidden_layers=4, num_attention_heads=4, intermediate_size=512,
                      max_position_embeddings=130)


def teacher_spread_case():
    """The DISCRIMINATING teacher fixture: 40 pair sequences of distinct random tokens, ragged lengths,
    weights of the "spread" recipe (trained-like gains), so the fp32 logits spread over several units and their ORDER
    is something a kernel has to get right."""
    from semantic_search_kd_amd.teacher import TeacherConfig, synthetic_pair_token_ids, synthetic_teacher_state_dict

    cfg = TeacherConfig(**TEACHER_SPREAD)
    sd = synthetic_teacher_state_dict(cfg, recipe="spread")
    ids, mask = synthetic_pair_token_ids(cfg, 40, 96, seed=53)
    return cfg, sd, ids, mask


def make_xlmr_spread():
    from oracle import teacher as teacher_oracle

    cfg, sd, ids, mask = teacher_spread_case()
    want = _xlmr_logits(cfg, sd, ids, mask)
    got = teacher_oracle.logits(sd, ids, mask, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.layer_norm_eps, cfg.pad_token_id)
    spread = float(want.max() - want.min())
    gaps = np.diff(np.sort(want))
    print(f"[xlmr_spread] oracle vs transformers: max |diff| = {np.abs(got - want).max():.3e}; {len(want)} logits, spread "
          f"{spread:.3f}, std {want.std():.3f}, smallest gap between neighbours {gaps.min():.4f}")
    assert np.abs(got - want).max() < 2e-5 and spread >= 2.0 and len(want) >= 32
    np.savez_compressed(HERE / "xlmr_spread.npz", input_ids=ids, attention_mask=mask, logits=want)


def make_kd_loss():
    """Losses and autograd gradients from the REFERENCE'S OWN code (src/kd/losses.py), imported from
    /root/reference in this container only.  Its one missing dependency is a logging package used
    for ``logger.info`` lines; an in-process no-op stand-in is registered for it (it takes no part
    in the arithmetic).  The oracle restatement is asserted equal here."""
    import types

    from oracle import kd_losses as kd_oracle

    sys.path.insert(0, "/root/reference")
    stub = types.ModuleType("loguru")

    class _Silent:
        def __getattr__(self, name):
            return lambda *a, **k: None

    stub.logger = _Silent()
    sys.modules.setdefault("loguru", stub)
    from src.kd.losses import CombinedKDLoss, ContrastiveLoss, ListwiseKDLoss, MarginMSELoss

    out = {}
    cases = {"b4": (4, 9, 7), "b64": (64, 9, 8), "b5d33": (5, 33, 9), "ties": (3, 9, 10)}
    for name, (b, d, seed) in cases.items():
        g = torch.Generator().manual_seed(seed)
        s = torch.randn(b, d, generator=g)
        t = torch.randn(b, d, generator=g) * 3.0
        if name == "ties":  # repeated row maxima: which index receives the max's gradient matters
            s[0, 2] = s[0, 5] = s[0].max() + 0.5
            t[1, 0] = t[1, 8] = t[1].max() + 1.0
        out[f"{name}_s"], out[f"{name}_t"] = s.numpy(), t.numpy()
        for temp in (4.0, 3.0, 2.0):
            tag = f"{name}_T{int(temp)}"
            comps = {"mm": MarginMSELoss(temp), "lk": ListwiseKDLoss(temp), "c": ContrastiveLoss(0.05)}
            for key, fn in comps.items():
                sv = s.clone().requires_grad_(True)
                loss = fn(sv, t) if key != "c" else fn(sv)
                loss.backward()
                out[f"{tag}_{key}"] = np.float64(loss.item())
                out[f"{tag}_{key}_grad"] = sv.grad.numpy().copy()
            comb = CombinedKDLoss()
            comb.update_temperature((4.0 - temp) / 2.0)  # 4 -> 2 linear annealing
            assert abs(comb.current_temperature - temp) < 1e-12
            sv = s.clone().requires_grad_(True)
            res = comb(sv, t)
            res["loss"].backward()
            out[f"{tag}_total"] = np.float64(res["loss"].item())
            out[f"{tag}_total_grad"] = sv.grad.numpy().copy()
            # the restatement agrees with the reference
            o, og = kd_oracle.combined(s.numpy(), t.numpy(), temp)
            assert abs(o["loss"] - res["loss"].item()) < 2e-5 * max(1.0, abs(o["loss"]))
            for key, ref in (("margin_mse", res["margin_mse"]), ("listwise_kd", res["listwise_kd"]),
                             ("contrastive", res["contrastive"])):
                assert abs(o[key] - ref) < 2e-5 * max(1.0, abs(ref)), (tag, key, o[key], ref)
            assert np.abs(og - sv.grad.numpy()).max() < 2e-5 * max(1.0, np.abs(og).max()), tag
    np.savez_compressed(HERE / "kd_loss.npz", **out)


API_MODELS = ("SearchRequest", "SearchResult", "SearchResponse", "EncodeRequest", "EncodeResponse",
              "HealthResponse", "ErrorResponse")


def contract_of(model) -> dict:
    """The validation-relevant part of a pydantic model's JSON schema: field names, types, defaults,
    bounds, required fields and references - without titles, descriptions or examples."""
    drop = {"title", "description", "example", "examples"}

    def strip(node):
        if isinstance(node, dict):
            return {k: strip(v) for k, v in sorted(node.items()) if k not in drop}
        if isinstance(node, list):
            return [strip(v) for v in node]
        return node

    return strip(model.model_json_schema())


def make_api_schemas():
    """Request / response contract of the reference's service (src/serve/schemas.py, importable
    here as is): the boundary the drop-in must keep."""
    import json

    sys.path.insert(0, "/root/reference")
    from src.serve import schemas as ref

    out = {name: contract_of(getattr(ref, name)) for name in API_MODELS}
    (HERE / "api_schemas.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    steps = {
        "search_small": make_search_small, "search_1k": make_search_1k, "search_ref": make_search_ref,
        "pool_norm": make_pool_norm, "bert_l2": lambda: make_bert(2, "l2"), "bert_l12": lambda: make_bert(12, "l12"),
        "bert_stress_l2": lambda: make_bert(2, "stress_l2", stress=True),
        "bert_stress_l12": lambda: make_bert(12, "stress_l12", stress=True), "bert_grads": make_bert_grads,
        "xlmr": make_xlmr, "xlmr_spread": make_xlmr_spread, "kd_loss": make_kd_loss, "ance": make_ance,
        "teacher_mining": make_teacher_mining, "api_schemas": make_api_schemas,
    }
    wanted = sys.argv[1:] or list(steps)   # ``python make_golden.py xlmr_spread`` regenerates one fixture
    for name in wanted:
        steps[name]()
    for p in sorted(HERE.glob("*.npz")):
        print(f"{p.name}: {p.stat().st_size / 1024:.1f} KiB")
