"""Teacher cross-encoder (BASELINE cfg 5 model) on the GPU vs the oracle and the committed
``transformers.XLMRobertaForSequenceClassification`` logits.

The score-level gate is ``test_teacher_spread_fixture_scores_and_order`` (+ the 24-layer test): 40 inputs whose
fp32 logits spread over 2.7 units; the HIP logits must sit within 3 % of that spread AND reproduce the order
(Kendall tau >= 0.95, same top-5 set, same top-1).  A reranker's product is an ordering (reference:
src/serve/app.py:321-339, src/mining/miners.py:140-151).

``xlmr_small`` is the round-1 fixture: random-INIT weights and near-identical inputs (its token recipe clamps the
800-word vocabulary to one id), so its five logits lie within 0.067 of each other behind a head that is 32x
steeper than a trained one - a sensitivity test, not an ordering test.  Where its error comes from was measured,
not estimated (tools/teacher_head_bisect.py, gpurun_out/r03_head_bisect.log): an op-by-op bf16 emulation of the
round-2 head on the HIP <s> state reproduces the HIP logits BIT FOR BIT (no head defect); an fp32 head on the same
state is off by 0.0385; the HIP <s> state differs from the oracle's by 0.5-0.65 % (cosine 0.99998), i.e. three
layers of bf16 activations / weights; rounding the ORACLE's <s> state once to bf16 already moves a logit by 0.007.
The head now runs in fp32 (``teacher_head_kernel``) and this fixture keeps the 0.05 gate of the round-2 test before
it was loosened.
"""
LOGIT_ATOL = 0.05          # xlmr_small / random-init weights behind a steep head: see above (measured 0.0385)
SPREAD_REL_TOL = 0.03      # |d logit| <= 3 % of the fixture's logit spread
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import encoder as enc_oracle
from oracle import teacher as teacher_oracle


def _case(cfg, B, S, lengths, seed):
    ids, mask = enc_oracle.synthetic_token_ids(B, S, seed=seed, vocab=cfg.vocab_size, lengths=lengths)
    ids = np.where(mask == 1, np.maximum(ids, 4), cfg.pad_token_id).astype(np.int32)
    ids[:, 0] = 0
    return ids, mask


def test_teacher_oracle_matches_transformers_fixture():
    """CPU: oracle/teacher.py reproduces the committed XLMRobertaForSequenceClassification logits."""
    import sys

    sys.path.insert(0, str(GOLDEN))
    from make_golden import teacher_case

    gold = np.load(GOLDEN / "xlmr_small.npz")
    cfg, sd, ids, mask = teacher_case()
    assert np.array_equal(ids, gold["input_ids"])
    got = teacher_oracle.logits(sd, ids, mask, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.layer_norm_eps, cfg.pad_token_id)
    np.testing.assert_allclose(got, gold["logits"], atol=1e-5)


def test_teacher_oracle_matches_spread_fixture():
    """CPU: the oracle reproduces the committed logits of the DISCRIMINATING fixture (40 inputs, spread 2.7)."""
    import sys

    sys.path.insert(0, str(GOLDEN))
    from make_golden import teacher_spread_case

    gold = np.load(GOLDEN / "xlmr_spread.npz")
    cfg, sd, ids, mask = teacher_spread_case()
    assert np.array_equal(ids, gold["input_ids"]) and np.array_equal(mask, gold["attention_mask"])
    assert len(gold["logits"]) >= 32 and gold["logits"].max() - gold["logits"].min() >= 2.0
    got = teacher_oracle.logits(sd, ids, mask, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.layer_norm_eps, cfg.pad_token_id)
    np.testing.assert_allclose(got, gold["logits"], atol=2e-5)


@pytest.mark.gpu
def test_teacher_small_matches_fixture_and_oracle(gpu):
    import sys

    sys.path.insert(0, str(GOLDEN))
    from make_golden import teacher_case

    from semantic_search_kd_amd import TeacherModel

    gold = np.load(GOLDEN / "xlmr_small.npz")
    cfg, sd, ids, mask = teacher_case()
    teacher = TeacherModel("synthetic", "cuda:0", config=cfg, state_dict=sd)
    got = teacher.score_token_ids(ids, mask).cpu().numpy()
    assert np.abs(got - gold["logits"]).max() <= LOGIT_ATOL, (got, gold["logits"])
    # batch-mate independence and determinism
    alone = teacher.score_token_ids(ids[2:3, :33], mask[2:3, :33]).cpu().numpy()
    assert abs(alone[0] - got[2]) <= LOGIT_ATOL
    assert np.array_equal(teacher.score_token_ids(ids, mask).cpu().numpy(), got)
    assert 0.0 < TeacherModel.get_confidence(got[0]) < 1.0 and TeacherModel.get_confidence(0.0) == 0.5


def _kendall_tau(a, b):
    from scipy.stats import kendalltau

    return float(kendalltau(a, b).statistic)


@pytest.mark.gpu
def test_teacher_spread_fixture_scores_and_order(gpu):
    """The discriminating fixture (committed ``XLMRobertaForSequenceClassification`` logits of 40 distinct inputs,
    spread 2.7): HIP scores within 3 % of the spread, and the ORDER survives - Kendall tau >= 0.95, identical top-5
    set, identical top-1.  A constant output, a transposed head or a dropped layer cannot pass."""
    import sys

    sys.path.insert(0, str(GOLDEN))
    from make_golden import teacher_spread_case

    from semantic_search_kd_amd import TeacherModel

    gold = np.load(GOLDEN / "xlmr_spread.npz")["logits"]
    cfg, sd, ids, mask = teacher_spread_case()
    teacher = TeacherModel("synthetic", "cuda:0", config=cfg, state_dict=sd)
    got = teacher.score_token_ids(ids, mask).cpu().numpy()
    spread = float(gold.max() - gold.min())
    err = np.abs(got - gold).max()
    tau = _kendall_tau(got, gold)
    print(f"spread fixture: max |d| = {err:.4f} = {100 * err / spread:.2f} % of the spread {spread:.3f}; tau = {tau:.4f}")
    assert spread >= 2.0 and len(gold) >= 32
    assert err <= SPREAD_REL_TOL * spread, (err, spread)
    assert tau >= 0.95, tau
    assert set(np.argsort(-got)[:5]) == set(np.argsort(-gold)[:5]) and got.argmax() == gold.argmax()
    # one pair at a time (its own launch, its own padding): same scores within the gate, same order
    alone = np.array([float(teacher.score_token_ids(ids[i : i + 1, : int(mask[i].sum())], mask[i : i + 1, : int(mask[i].sum())])[0])
                      for i in range(len(gold))])
    assert np.abs(alone - gold).max() <= SPREAD_REL_TOL * spread and _kendall_tau(alone, gold) >= 0.95


@pytest.mark.gpu
def test_teacher_full_depth_24_layers_vs_oracle(gpu):
    """BASELINE cfg 5 at its REAL depth: XLM-R-large (24 layers x hidden 1024, 16 heads, FFN 4096; vocabulary cut to
    3 000 rows) with trained-like gains, 12 pairs x 64 tokens, against the fp32 oracle - final hidden states of every
    real token, the logits and their order.  (bench.py runs this depth at 128 x 256 but can only check ``finite``.)"""
    from semantic_search_kd_amd import TeacherConfig, TeacherModel, _native
    from semantic_search_kd_amd.teacher import synthetic_pair_token_ids, synthetic_teacher_state_dict

    cfg = TeacherConfig(vocab_size=3000, max_position_embeddings=200)
    assert cfg.num_hidden_layers == 24 and cfg.hidden_size == 1024
    sd = synthetic_teacher_state_dict(cfg, recipe="spread")
    B, S = 12, 64
    ids, mask = synthetic_pair_token_ids(cfg, B, S, seed=71)
    teacher = TeacherModel("synthetic", "cuda:0", config=cfg, state_dict=sd)
    got = teacher.score_token_ids(ids, mask).cpu().numpy()
    t = {k: torch.from_numpy(v) for k, v in sd.items()}
    hs = enc_oracle.bert_hidden_states_torch(t, ids, mask, 24, 16, cfg.layer_norm_eps, pos_offset=cfg.pad_token_id + 1)[-1]
    want = (torch.tanh(hs[:, 0] @ t["classifier.dense.weight"].T + t["classifier.dense.bias"])
            @ t["classifier.out_proj.weight"].T + t["classifier.out_proj.bias"])[:, 0].numpy()
    lib = _native.load()
    out = torch.empty((B, S, 1024), dtype=torch.bfloat16, device="cuda")
    d_ids, d_mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    ws = torch.empty(int(lib.sskd_generic_workspace_bytes(teacher._cfg, B, S, 0)), dtype=torch.uint8, device="cuda")
    _native.check(lib.sskd_generic_forward(teacher._cfg, teacher._w, d_ids.data_ptr(), d_mask.data_ptr(), B, S, 0, 0, 0,
                                           out.data_ptr(), ws.data_ptr(), ws.numel(), int(torch.cuda.current_stream().cuda_stream)))
    h, ref, m = out.float().cpu().numpy(), hs.numpy(), mask.astype(bool)
    cos = (h[m] * ref[m]).sum(1) / (np.linalg.norm(h[m], axis=1) * np.linalg.norm(ref[m], axis=1))
    spread = float(want.max() - want.min())
    err = np.abs(got - want).max()
    tau = _kendall_tau(got, want)
    print(f"24 layers: min cos {cos.min():.5f}; logits max |d| = {err:.4f} = {100 * err / spread:.2f} % of the spread "
          f"{spread:.3f}; tau = {tau:.3f}")
    assert cos.min() >= 0.999, cos.min()   # measured 0.99976 (gpurun_out/r03_t2.log): the 2-layer gate holds at depth 24
    assert spread >= 1.5 and err <= SPREAD_REL_TOL * spread, (err, spread)
    # pairs whose fp32 logits differ by more than twice the observed error must keep their order
    for i in range(B):
        for j in range(B):
            if want[i] - want[j] > 2 * SPREAD_REL_TOL * spread:
                assert got[i] > got[j], (i, j, want[i], want[j], got[i], got[j])
    assert tau >= 0.9


@pytest.mark.gpu
def test_teacher_large_shape_layers_vs_oracle(gpu):
    """XLM-R-large layer shape (hidden 1024, 16 heads x 64, FFN 4096), 2 layers, reduced vocabulary."""
    from semantic_search_kd_amd import TeacherConfig, TeacherModel, _native
    from semantic_search_kd_amd.teacher import synthetic_teacher_state_dict

    cfg = TeacherConfig(vocab_size=3000, num_hidden_layers=2, max_position_embeddings=200)
    sd = synthetic_teacher_state_dict(cfg)
    ids, mask = _case(cfg, 4, 128, [128, 97, 40, 6], seed=61)
    teacher = TeacherModel("synthetic", "cuda:0", config=cfg, state_dict=sd)
    got = teacher.score_token_ids(ids, mask).cpu().numpy()
    want = teacher_oracle.logits(sd, ids, mask, 2, 16, cfg.layer_norm_eps, cfg.pad_token_id)
    assert np.abs(got - want).max() <= LOGIT_ATOL, (got, want)
    # final hidden states through the C-ABI (pool = 0) vs the oracle
    lib = _native.load()
    out = torch.empty((4, 128, 1024), dtype=torch.bfloat16, device="cuda")
    d_ids, d_mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    ws = torch.empty(int(lib.sskd_generic_workspace_bytes(teacher._cfg, 4, 128, 0)), dtype=torch.uint8, device="cuda")
    _native.check(lib.sskd_generic_forward(teacher._cfg, teacher._w, d_ids.data_ptr(), d_mask.data_ptr(), 4, 128, 0, 0, 0,
                                           out.data_ptr(), ws.data_ptr(), ws.numel(), int(torch.cuda.current_stream().cuda_stream)))
    t = {k: torch.from_numpy(v) for k, v in sd.items()}
    ref = enc_oracle.bert_hidden_states_torch(t, ids, mask, 2, 16, cfg.layer_norm_eps, pos_offset=2)[-1].numpy()
    h = out.float().cpu().numpy()
    m = mask.astype(bool)
    cos = (h[m] * ref[m]).sum(1) / (np.linalg.norm(h[m], axis=1) * np.linalg.norm(ref[m], axis=1))
    assert cos.min() >= 0.999


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,heads,S,lengths", [
    (256, 2, 96, [96, 61, 33, 5]),        # head width 128, three key tiles, one workgroup per (row, head)
    (128, 4, 512, [512, 300, 257, 32]),   # head width 32, 16 query tiles = two workgroups per (row, head)
    (128, 2, 288, [288, 200, 31]),        # head width 64, nine tiles: the second workgroup has one active wave
])
def test_inference_forward_fused_attention_shapes(gpu, hidden, heads, S, lengths):
    """The inference forward runs attention as one fused kernel (no score matrix) and GELU inside FFN1's
    epilogue: hold the hidden states of every real token to the oracle across head widths, tile counts and
    ragged key masks (rows with masked tails exercise the skipped key tiles and the additive bias)."""
    from semantic_search_kd_amd import TeacherConfig, TeacherModel, _native
    from semantic_search_kd_amd.teacher import synthetic_teacher_state_dict

    cfg = TeacherConfig(vocab_size=900, hidden_size=hidden, num_hidden_layers=2, num_attention_heads=heads,
                        intermediate_size=2 * hidden, max_position_embeddings=S + 4)
    sd = synthetic_teacher_state_dict(cfg)
    B = len(lengths)
    ids, mask = _case(cfg, B, S, lengths, seed=S + hidden)
    teacher = TeacherModel("synthetic", "cuda:0", config=cfg, state_dict=sd)
    lib = _native.load()
    out = torch.empty((B, S, hidden), dtype=torch.bfloat16, device="cuda")
    d_ids, d_mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    ws = torch.empty(int(lib.sskd_generic_workspace_bytes(teacher._cfg, B, S, 0)), dtype=torch.uint8, device="cuda")
    _native.check(lib.sskd_generic_forward(teacher._cfg, teacher._w, d_ids.data_ptr(), d_mask.data_ptr(), B, S, 0, 0, 0,
                                           out.data_ptr(), ws.data_ptr(), ws.numel(), int(torch.cuda.current_stream().cuda_stream)))
    t = {k: torch.from_numpy(v) for k, v in sd.items()}
    ref = enc_oracle.bert_hidden_states_torch(t, ids, mask, 2, heads, cfg.layer_norm_eps, pos_offset=2)[-1].numpy()
    h = out.float().cpu().numpy()
    m = mask.astype(bool)
    assert np.isfinite(h).all()
    cos = (h[m] * ref[m]).sum(1) / (np.linalg.norm(h[m], axis=1) * np.linalg.norm(ref[m], axis=1))
    assert cos.min() >= 0.999, cos.min()
    # the training-mode forward (unfused, activations saved) must agree with the fused one
    out_t = torch.empty_like(out)
    ws_t = torch.empty(int(lib.sskd_generic_workspace_bytes(teacher._cfg, B, S, 1)), dtype=torch.uint8, device="cuda")
    _native.check(lib.sskd_generic_forward(teacher._cfg, teacher._w, d_ids.data_ptr(), d_mask.data_ptr(), B, S, 1, 0, 0,
                                           out_t.data_ptr(), ws_t.data_ptr(), ws_t.numel(), int(torch.cuda.current_stream().cuda_stream)))
    ht = out_t.float().cpu().numpy()
    cos_t = (h[m] * ht[m]).sum(1) / (np.linalg.norm(h[m], axis=1) * np.linalg.norm(ht[m], axis=1))
    assert cos_t.min() >= 0.9995, cos_t.min()


@pytest.mark.gpu
def test_teacher_score_pairs_api_and_rerank_route(gpu):
    """score(pairs) (lists or tuples, any batch_size), predict / predict_score aliases, and the /search
    rerank branch (reference: src/serve/app.py:321-339) driving the real teacher object."""
    from semantic_search_kd_amd import TeacherConfig, TeacherModel
    from semantic_search_kd_amd.bench_support import synthetic_passages, synthetic_vocab
    from semantic_search_kd_amd.encoder import build_wordpiece_tokenizer
    from semantic_search_kd_amd.teacher import synthetic_teacher_state_dict

    vocab = synthetic_vocab(4000)
    cfg = TeacherConfig(vocab_size=4000, hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                        intermediate_size=256, max_position_embeddings=258)
    sd = synthetic_teacher_state_dict(cfg, recipe="spread")
    teacher = TeacherModel("synthetic", "cuda:0", config=cfg, state_dict=sd, tokenizer=build_wordpiece_tokenizer(vocab))
    docs = synthetic_passages(vocab, 40, seed=4)
    pairs = [("bure didi fusa", d) for d in docs[:20]] + [["majini horo", d] for d in docs[20:]]
    scores = teacher.score(pairs, batch_size=7)
    assert len(scores) == 40 and all(isinstance(s, float) for s in scores)
    ids, mask = teacher.tokenize_pairs(pairs)
    want = teacher_oracle.logits(sd, ids, mask, 2, 4, cfg.layer_norm_eps, cfg.pad_token_id)
    spread = float(want.max() - want.min())
    assert spread >= 1.0, spread
    assert np.abs(np.array(scores) - want).max() <= SPREAD_REL_TOL * spread
    assert _kendall_tau(np.array(scores), want) >= 0.9 and int(np.argmax(scores)) == int(want.argmax())
    assert abs(teacher.predict_score(*pairs[3]) - scores[3]) <= LOGIT_ATOL
    assert np.allclose(teacher.predict(pairs[:5]), scores[:5], atol=LOGIT_ATOL)
    assert teacher.score([]) == []
    with pytest.raises(FileNotFoundError, match="never downloads"):
        TeacherModel("BAAI/bge-reranker-large", device="cuda:0")


def _dp_teacher(device="cuda:0"):
    from semantic_search_kd_amd import TeacherConfig, TeacherModel
    from semantic_search_kd_amd.bench_support import synthetic_passages, synthetic_vocab
    from semantic_search_kd_amd.encoder import build_wordpiece_tokenizer
    from semantic_search_kd_amd.teacher import synthetic_teacher_state_dict

    vocab = synthetic_vocab(4000)
    cfg = TeacherConfig(vocab_size=4000, hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                        intermediate_size=256, max_position_embeddings=258)
    teacher = TeacherModel("synthetic", device, config=cfg, state_dict=synthetic_teacher_state_dict(cfg, recipe="spread"),
                           tokenizer=build_wordpiece_tokenizer(vocab))
    docs = synthetic_passages(vocab, 61, seed=9)
    queries = ["bure didi fusa", "majini horo", "kalo seti wanu bure"]
    return teacher, queries, docs


def _dp_worker(rank, world, port, out_dir):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    multi = torch.cuda.device_count() >= world
    dev = f"cuda:{rank if multi else 0}"
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if multi else "gloo", rank=rank, world_size=world,
                            **({"device_id": torch.device(dev)} if multi else {}))
    try:
        from semantic_search_kd_amd import TeacherMiner

        teacher, queries, docs = _dp_teacher(dev)
        teacher.data_parallel()
        pairs = [(queries[i % 3], d) for i, d in enumerate(docs)]          # 61 pairs: ragged shards (31 + 30)
        scores = teacher.score(pairs)
        texts = {f"d{i}": d for i, d in enumerate(docs)}
        cands = [[f"d{i}" for i in range(0, 25)], [], [f"d{i}" for i in range(20, 61)] + ["missing"]]
        ids, kept = TeacherMiner(teacher, confidence_threshold=0.3).mine(queries, cands, texts, top_k=6)
        np.savez(os.path.join(out_dir, f"dp{rank}.npz"), scores=np.array(scores), ids=np.array([",".join(x) for x in ids]),
                 kept=np.concatenate([np.array(k, np.float64) for k in kept]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_teacher_data_parallel_two_ranks_and_teacher_miner(gpu, tmp_path):
    """cfg 5's parallelism: ``TeacherModel.data_parallel()`` splits the pairs over a 2-rank process group (contiguous
    ranges, one all-gather as the final concat; RCCL when 2 GPUs are visible, else both ranks share cuda:0 over gloo);
    every rank must return what one process returns.  ``TeacherMiner`` (reference: src/mining/miners.py:104-158) runs
    on top of it with the real teacher object."""
    import socket

    import torch.multiprocessing as mp

    from semantic_search_kd_amd import TeacherMiner

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    teacher, queries, docs = _dp_teacher()
    pairs = [(queries[i % 3], d) for i, d in enumerate(docs)]
    want = np.array(teacher.score(pairs))
    texts = {f"d{i}": d for i, d in enumerate(docs)}
    cands = [[f"d{i}" for i in range(0, 25)], [], [f"d{i}" for i in range(20, 61)] + ["missing"]]
    ids, kept = TeacherMiner(teacher, confidence_threshold=0.3).mine(queries, cands, texts, top_k=6)
    assert [len(x) for x in ids][1] == 0 and all(len(x) <= 6 for x in ids)
    spread = float(want.max() - want.min())
    got = [np.load(tmp_path / f"dp{r}.npz") for r in range(2)]
    for g in got:
        # a pair's batch-mates differ between 1 and 2 ranks (launch padding): same scores within the parity gate
        assert np.abs(g["scores"] - want).max() <= SPREAD_REL_TOL * spread
        assert _kendall_tau(g["scores"], want) >= 0.97
    # both ranks hold the SAME gathered scores, hence the same mined ids and kept scores
    assert np.array_equal(got[0]["scores"], got[1]["scores"]) and list(got[0]["ids"]) == list(got[1]["ids"])
    assert np.array_equal(got[0]["kept"], got[1]["kept"])
    if np.array_equal(got[0]["scores"], want):
        assert list(got[0]["ids"]) == [",".join(x) for x in ids]


@pytest.mark.gpu
def test_teacher_two_branch_batch_equals_its_halves(gpu):
    """sskd_teacher_score runs batches of >= 2 x 16 384 tokens as two halves on two streams (csrc/train.hip, one of them a
    side stream forked from / joined into the caller's).  A pair's score does not depend on its batch-mates: 128 pairs x
    256 tokens must equal the two 64-pair calls (each below the split threshold: one stream) bit for bit."""
    from semantic_search_kd_amd import TeacherModel
    from semantic_search_kd_amd.teacher import TeacherConfig, synthetic_teacher_state_dict

    cfg = TeacherConfig(vocab_size=800, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                        max_position_embeddings=260)
    teacher = TeacherModel("synthetic", "cuda:0", config=cfg, state_dict=synthetic_teacher_state_dict(cfg))
    rng = np.random.default_rng(3)
    P, S = 128, 256
    ids = rng.integers(4, cfg.vocab_size, size=(P, S)).astype(np.int32)
    ids[:, 0] = 0
    mask = np.ones((P, S), np.int32)
    for b in (1, 40, 64, 127):
        n = int(rng.integers(3, S))
        mask[b, n:] = 0
        ids[b, n:] = cfg.pad_token_id
    whole = teacher.score_token_ids(ids, mask).cpu().numpy()
    assert np.isfinite(whole).all()
    halves = np.concatenate([teacher.score_token_ids(ids[lo:lo + 64], mask[lo:lo + 64]).cpu().numpy() for lo in (0, 64)])
    assert np.array_equal(whole, halves)
