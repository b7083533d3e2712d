"""Sequence packing (varlen): ``encode_ragged`` / ``sskd_encoder_forward_packed`` against the fp32 oracle
run on every sequence ALONE, and against the padded path of the same library.

Tolerances: as tests/test_encoder_gpu.py for the oracle (cosine >= 0.999, |d| <= 4e-3).  Between the
packed and the padded HIP paths only the grouping of keys into 32-key tiles differs (the online
softmax sees another running maximum when it rounds the weights to bf16), so they agree far more
tightly than either does with fp32 - but not bit for bit: SURVEY.md section 8(f) allows 1e-2 for
bf16, we require 2e-3.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import encoder as enc_oracle
from semantic_search_kd_amd import BertConfig, Mi355xSentenceEncoder, _native, synthetic_state_dict
from semantic_search_kd_amd.bench_support import marco_like_lengths

COS_MIN = 0.999
EMB_ATOL = 4e-3


def _cos(a, b):
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


def _ragged(lengths, seed, vocab=30522):
    g = np.random.Generator(np.random.PCG64(seed))
    seqs = []
    for n in lengths:
        ids = g.integers(999, vocab, size=int(n), dtype=np.int64).astype(np.int32)
        ids[0] = 101
        if n > 1:
            ids[-1] = 102
        seqs.append(ids)
    return np.concatenate(seqs), np.asarray(lengths, np.int32), seqs


def _oracle_each(sd, seqs, layers):
    return np.stack([enc_oracle.encode_token_ids(sd, s[None, :], None, layers)[0] for s in seqs])


def _padded(enc, seqs):
    width = max(len(s) for s in seqs)
    ids = np.zeros((len(seqs), width), np.int32)
    mask = np.zeros((len(seqs), width), np.int32)
    for i, s in enumerate(seqs):
        ids[i, : len(s)] = s
        mask[i, : len(s)] = 1
    return enc.encode_token_ids(ids, mask).cpu().numpy()


def test_pack_plan_is_a_valid_tight_placement(native_lib):
    """Host planner (no GPU): every sequence placed once, no overlap, < 3 % padding on MS MARCO-shaped lengths."""
    lens = marco_like_lengths(3000, seed=3).astype(np.int32)
    table = np.empty((lens.size, 4), np.int32)
    rows = ctypes.c_int()
    assert native_lib.sskd_pack_plan(lens.ctypes.data, lens.size, 256, table.ctypes.data, rows) == 0
    occ = np.zeros((rows.value, 256), np.int32)
    for r, lo, hi, i in table:
        occ[r, lo:hi] += 1
    assert occ.max() == 1 and np.array_equal(table[:, 2] - table[:, 1], lens)
    assert np.array_equal(table[:, 3], np.arange(lens.size))
    assert rows.value * 256 / lens.sum() - 1.0 < 0.03
    bad = np.array([300], np.int32)
    assert native_lib.sskd_pack_plan(bad.ctypes.data, 1, 256, table.ctypes.data, rows) == 1  # longer than a row
    assert native_lib.sskd_pack_plan(lens.ctypes.data, 0, 256, table.ctypes.data, rows) == 0 and rows.value == 0


@pytest.fixture(scope="module")
def enc_l2(gpu):
    cfg = BertConfig(num_hidden_layers=2)
    return Mi355xSentenceEncoder.from_synthetic(cfg, device="cuda:0"), cfg, synthetic_state_dict(cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("lengths", [
    [9, 3, 12, 7],                      # one 32-token row, four segments in one tile
    [1],                                # single [CLS]
    [256],                              # exactly one full row
    [256, 1, 255, 2, 129, 127, 64, 64, 64, 64, 33, 31],
    [5] * 120,                          # query-like: ~50 segments per row, 6-7 per tile
    [40, 75, 110, 68, 68, 91, 23, 256, 180, 12, 77, 54, 99, 130, 61, 8],
])
def test_packed_matches_oracle_and_padded_path(enc_l2, lengths):
    enc, cfg, sd = enc_l2
    flat, lens, seqs = _ragged(lengths, seed=len(lengths) * 7 + lengths[0])
    emb = enc.encode_ragged(flat, lens).cpu().numpy()
    assert emb.shape == (len(lengths), 384)
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    want = _oracle_each(sd, seqs, cfg.num_hidden_layers)
    assert _cos(emb, want).min() >= COS_MIN
    assert np.abs(emb - want).max() <= EMB_ATOL
    pad = _padded(enc, seqs)
    assert np.abs(emb - pad).max() <= 2e-3 and _cos(emb, pad).min() >= 0.99999
    # deterministic, and independent of the order the sequences arrive in (other rows / offsets)
    assert np.array_equal(enc.encode_ragged(flat, lens).cpu().numpy(), emb)
    perm = np.random.default_rng(1).permutation(len(lengths))
    flat_p = np.concatenate([seqs[i] for i in perm])
    emb_p = enc.encode_ragged(flat_p, lens[perm]).cpu().numpy()
    assert np.abs(emb_p - emb[perm]).max() <= 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("stress", [False, True])
def test_packed_marco_shaped_batch_12_layers(gpu, stress):
    """Several launches' worth of MS MARCO-shaped lengths through the 12-layer model (benign and
    hard-case weights): < 5 % padding, sampled sequences vs the oracle."""
    cfg = BertConfig()
    enc = Mi355xSentenceEncoder.from_synthetic(cfg, device="cuda:0", stress=stress)
    sd = synthetic_state_dict(cfg, stress=stress)
    lengths = marco_like_lengths(4000, seed=5)
    flat, lens, seqs = _ragged(lengths, seed=99)
    emb = enc.encode_ragged(flat, lens).cpu().numpy()
    assert np.isfinite(emb).all()
    assert enc.last_encode_stats["padding_overhead"] < 0.05
    rows = [0, 1, 777, 1699, 1700, 2500, 3999, int(np.argmax(lengths)), int(np.argmin(lengths))]
    want = _oracle_each(sd, [seqs[i] for i in rows], cfg.num_hidden_layers)
    cos = _cos(emb[rows], want)
    assert cos.min() >= COS_MIN, cos
    assert np.abs(emb[rows] - want).max() <= EMB_ATOL


@pytest.mark.gpu
def test_packed_rejects_bad_input(enc_l2):
    enc, _, _ = enc_l2
    with pytest.raises(ValueError, match="lengths must lie"):
        enc.encode_ragged(np.zeros(300, np.int32), np.array([300], np.int32))
    with pytest.raises(ValueError, match="sum"):
        enc.encode_ragged(np.zeros(5, np.int32), np.array([3, 3], np.int32))
    assert enc.encode_ragged(np.zeros(0, np.int32), np.zeros(0, np.int32)).shape == (0, 384)


@pytest.mark.gpu
def test_ance_refresh_and_mine_on_gpu(gpu):
    """ANCE refresh composition (reference: src/mining/miners.py:184-253 + docs/adr-003): re-encode the
    corpus with the current student (packed varlen encoder, C++ tokenizer), rebuild the exact index in
    HBM, search, apply the margin rule - checked against the same rule applied to oracle-side numpy
    products of the very embeddings the GPU produced."""
    from semantic_search_kd_amd import ANCEMiner, StudentModel
    from semantic_search_kd_amd.bench_support import synthetic_passages, synthetic_vocab
    from semantic_search_kd_amd.encoder import build_wordpiece_tokenizer
    from semantic_search_kd_amd.mining import select_adversarial

    vocab = synthetic_vocab()
    enc = Mi355xSentenceEncoder.from_synthetic(BertConfig(num_hidden_layers=2), device="cuda:0",
                                               tokenizer=build_wordpiece_tokenizer(vocab))
    student = StudentModel.from_encoder(enc, "e5-small-v2-synthetic")
    corpus = synthetic_passages(vocab, 600, seed=21)
    ids = [f"doc{i}" for i in range(len(corpus))]
    queries = [" ".join(c.split()[:6]) for c in corpus[:20]]      # each query is the head of a passage
    positives = [[ids[i]] for i in range(20)]
    miner = ANCEMiner(student, margin=0.05)
    index = miner.refresh(ids, corpus)
    assert index.ntotal == 600
    got = miner.mine_from_index(queries, positives, top_k=5, search_k=50)
    d = student.encode_documents(corpus)
    q = student.encode_queries(queries)
    sims = q.astype(np.float64) @ d.astype(np.float64).T
    for qi in range(20):
        order = [j for j in np.argsort(-sims[qi], kind="stable")[:50] if j != qi]
        want = select_adversarial([ids[j] for j in order], sims[qi, order].astype(np.float32),
                                  sims[qi, [qi]].astype(np.float32), 0.05, 5)
        assert got[qi] == want
        assert ids[qi] not in got[qi]
    # the reference-shaped mine() over explicit candidate lists goes through the same student
    cands = [[ids[(7 * qi + j) % 600] for j in range(30)] for qi in range(20)]
    table = dict(zip(ids, corpus))
    mined = miner.mine(queries, positives, cands, table, table, top_k=3)
    for qi in range(20):
        cj = [(7 * qi + j) % 600 for j in range(30)]
        want = select_adversarial(cands[qi], sims[qi, cj].astype(np.float32), sims[qi, [qi]].astype(np.float32), 0.05, 3)
        assert set(mined[qi]) == set(want)
