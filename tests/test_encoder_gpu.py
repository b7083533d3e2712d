"""GPU parity of the bf16 encoder path against the fp32 oracle and the committed golden vectors.

Tolerances (floating point, stated here as the north star requires): the HIP path computes in
bf16 (8-bit mantissa) with fp32 accumulation, the reference in fp32.  For L2-normalised 384-d
embeddings we require cosine(e_hip, e_ref) >= 0.999 and max |e_hip - e_ref| <= 4e-3;
fp32 kernels (pool + normalise) must agree to 2e-6.
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import encoder as enc_oracle
from oracle import search as oracle
from semantic_search_kd_amd import BertConfig, Mi355xSentenceEncoder, StudentModel, _native, synthetic_state_dict
from semantic_search_kd_amd.weights import bf16_round, save_model_dir

pytestmark = pytest.mark.gpu

COS_MIN = 0.999
EMB_ATOL = 4e-3


def _stream():
    return int(torch.cuda.current_stream().cuda_stream)


def _cos(a, b):
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


@pytest.fixture(scope="module")
def enc_l2(gpu):
    cfg = BertConfig(num_hidden_layers=2)
    return Mi355xSentenceEncoder.from_synthetic(cfg, device="cuda:0"), cfg, synthetic_state_dict(cfg)


def test_pool_normalize_matches_golden(gpu, native_lib):
    gold = np.load(GOLDEN / "pool_norm.npz")
    g = np.random.Generator(np.random.PCG64(int(gold["seed"])))
    h = g.standard_normal((8, 64, 384), dtype=np.float32)
    mask = np.zeros((8, 64), np.int32)
    for b, n in enumerate(gold["lengths"]):
        mask[b, :n] = 1
    dh, dm = torch.from_numpy(h).cuda(), torch.from_numpy(mask).cuda()
    out = torch.empty((8, 384), device="cuda")
    for normalize, key in ((1, "normalized"), (0, "pooled")):
        _native.check(native_lib.sskd_pool_normalize(dh.data_ptr(), 0, dm.data_ptr(), 8, 64, normalize, out.data_ptr(), _stream()))
        np.testing.assert_allclose(out.cpu().numpy(), gold[key], rtol=0, atol=2e-6)
    # bf16 hidden input: same result on the bf16-rounded values
    hb = torch.from_numpy(h).cuda().to(torch.bfloat16)
    _native.check(native_lib.sskd_pool_normalize(hb.data_ptr(), 1, dm.data_ptr(), 8, 64, 1, out.data_ptr(), _stream()))
    want = oracle.pool_normalize(hb.float().cpu().numpy(), mask, True)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-6)
    # all-masked row: mean of nothing is 0 (clamp 1e-9), normalise leaves 0 (clamp 1e-12)
    dm0 = torch.zeros((8, 64), dtype=torch.int32, device="cuda")
    _native.check(native_lib.sskd_pool_normalize(dh.data_ptr(), 0, dm0.data_ptr(), 8, 64, 1, out.data_ptr(), _stream()))
    assert not out.cpu().numpy().any()


@pytest.mark.parametrize("tag", ["l2", "l12"])
def test_encoder_matches_transformers_golden(gpu, tag):
    """Embeddings vs the committed output of transformers.BertModel (fp32) on synthetic weights."""
    gold = np.load(GOLDEN / f"bert_{tag}.npz")
    cfg = BertConfig(num_hidden_layers=int(gold["layers"]))
    enc = Mi355xSentenceEncoder.from_synthetic(cfg, device="cuda:0")
    emb = enc.encode_token_ids(gold["input_ids"], gold["attention_mask"]).cpu().numpy()
    assert emb.shape == (4, 384) and emb.dtype == np.float32
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)  # test_model_validation.py:80-89
    cos = _cos(emb, gold["embeddings"])
    assert cos.min() >= COS_MIN, cos
    assert np.abs(emb - gold["embeddings"]).max() <= EMB_ATOL
    # CLS hidden state of the last layer (bf16 output vs fp32 golden)
    hs = enc.hidden_states(gold["input_ids"], gold["attention_mask"]).float().cpu().numpy()
    cls = hs[:, 0, :]
    assert _cos(cls, gold["last_hidden_cls"]).min() >= COS_MIN


@pytest.mark.parametrize("tag", ["stress_l2", "stress_l12"])
def test_encoder_stress_weights_match_transformers_golden(gpu, tag):
    """Hard-case weights (weights.synthetic_state_dict(stress=True): attention logits O(5) so
    softmax rows are peaky and the online-softmax rescale fires, LayerNorm gains in [0.3, 3],
    +-10 massive-activation channels), sequences of up to 7 key tiles; golden =
    transformers.BertModel fp32 (tests/golden/make_golden.py).  Same gate as the benign weights."""
    gold = np.load(GOLDEN / f"bert_{tag}.npz")
    cfg = BertConfig(num_hidden_layers=int(gold["layers"]))
    enc = Mi355xSentenceEncoder.from_synthetic(cfg, device="cuda:0", stress=True)
    emb = enc.encode_token_ids(gold["input_ids"], gold["attention_mask"]).cpu().numpy()
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    cos = _cos(emb, gold["embeddings"])
    assert cos.min() >= COS_MIN, cos
    assert np.abs(emb - gold["embeddings"]).max() <= EMB_ATOL
    hs = enc.hidden_states(gold["input_ids"], gold["attention_mask"]).float().cpu().numpy()
    assert _cos(hs[:, 0, :], gold["last_hidden_cls"]).min() >= COS_MIN


@pytest.mark.parametrize("stress", [False, True])
def test_encoder_cfg2_shape_sampled_rows_vs_oracle(gpu, stress):
    """BASELINE cfg 2 encoder shape (batch 512 x seq 256, 12 layers) on the GPU; 8 sampled rows are
    compared with the fp32 oracle run on those rows alone (valid because an embedding does not
    depend on its batch-mates: test_padding_invariance)."""
    cfg = BertConfig()
    enc = Mi355xSentenceEncoder.from_synthetic(cfg, device="cuda:0", stress=stress)
    sd = synthetic_state_dict(cfg, stress=stress)
    B, S = 512, 256
    rng = np.random.default_rng(77)
    lengths = [S] * B
    for b in (5, 100, 257, 511):      # a few ragged rows inside the otherwise full-length batch
        lengths[b] = int(rng.integers(3, S))
    ids, mask = enc_oracle.synthetic_token_ids(B, S, seed=2024, lengths=lengths)
    emb = enc.encode_token_ids(ids, mask).cpu().numpy()
    assert emb.shape == (B, 384) and np.isfinite(emb).all()
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    rows = np.array([0, 5, 100, 255, 256, 257, 300, 511])
    want = enc_oracle.encode_token_ids(sd, ids[rows], mask[rows], cfg.num_hidden_layers)
    cos = _cos(emb[rows], want)
    assert cos.min() >= COS_MIN, cos
    assert np.abs(emb[rows] - want).max() <= EMB_ATOL


def test_hidden_states_vs_oracle_same_weights(enc_l2):
    """Per-token hidden states vs the fp32 oracle run on the SAME bf16-rounded weights: isolates
    kernel arithmetic (bf16 activations) from weight quantisation."""
    enc, cfg, sd = enc_l2
    sd_b = {k: (bf16_round(v) if v.ndim == 2 else v) for k, v in sd.items()}
    ids, mask = enc_oracle.synthetic_token_ids(6, 80, seed=5, lengths=[80, 64, 33, 32, 31, 2])
    got = enc.hidden_states(ids, mask).float().cpu().numpy()
    want = enc_oracle.bert_hidden_states(sd_b, ids, mask, cfg.num_hidden_layers)
    m = mask.astype(bool)
    # LayerNorm output is O(1) per element; bf16 resolution there is 2^-8
    assert np.abs(got[m] - want[m]).max() < 6e-2
    assert np.abs(got[m] - want[m]).mean() < 6e-3
    assert _cos(got[m], want[m]).min() > 0.9995


@pytest.mark.parametrize("B,S,lengths", [
    (1, 1, None),            # [CLS]-only would be length 2; S=1 exercises the degenerate tile
    (1, 7, None),
    (3, 32, [32, 17, 2]),
    (2, 33, [33, 9]),
    (5, 100, [100, 77, 64, 3, 50]),
    (2, 256, [256, 130]),    # BASELINE cfg 2 sequence length
    (2, 300, [300, 257]),    # more than one 256-query block
    (1, 512, None),          # maximum length (config.py:29)
])
def test_encoder_shapes_and_ragged_masks(enc_l2, B, S, lengths):
    enc, cfg, sd = enc_l2
    ids, mask = enc_oracle.synthetic_token_ids(B, S, seed=100 + S, lengths=lengths)
    emb = enc.encode_token_ids(ids, mask).cpu().numpy()
    want = enc_oracle.encode_token_ids(sd, ids, mask, cfg.num_hidden_layers)
    assert _cos(emb, want).min() >= COS_MIN
    assert np.abs(emb - want).max() <= EMB_ATOL


@pytest.mark.parametrize("S", [32, 64, 96, 128, 160])
def test_short_sequences_packed_per_workgroup(enc_l2, S):
    """Sequences of <= 4 token tiles share an attention workgroup (8 / tiles of them): odd batch
    sizes, ragged lengths, and no dependence on which sequences share the workgroup."""
    enc, cfg, sd = enc_l2
    B = 19
    rng = np.random.default_rng(S)
    lengths = [int(x) for x in rng.integers(2, S + 1, size=B)]
    lengths[0] = S
    ids, mask = enc_oracle.synthetic_token_ids(B, S, seed=500 + S, lengths=lengths)
    emb = enc.encode_token_ids(ids, mask).cpu().numpy()
    want = enc_oracle.encode_token_ids(sd, ids, mask, cfg.num_hidden_layers)
    assert _cos(emb, want).min() >= COS_MIN
    assert np.abs(emb - want).max() <= EMB_ATOL
    # the same rows in another order / another batch land in other workgroup slots: identical bits
    perm = rng.permutation(B)
    emb_p = enc.encode_token_ids(ids[perm], mask[perm]).cpu().numpy()
    assert np.array_equal(emb_p, emb[perm])
    alone = enc.encode_token_ids(ids[5:6], mask[5:6]).cpu().numpy()
    assert np.array_equal(alone[0], emb[5])


def test_persistent_mlp_groups_match_one_group_per_workgroup(enc_l2):
    """The fused MLP is a persistent kernel: one workgroup per CU walks the 128-token groups blockIdx, blockIdx + gridDim,
    ... and its producers request the NEXT group's context rows while the consumers finish the current one
    (csrc/encoder.hip fused_mlp_ln_kernel).  A 328 x 256 batch is 656 groups: on 256 CUs 144 workgroups walk three groups,
    112 walk two (a ragged last round); the same rows in slices of 64 are at most 128 groups (one per workgroup, nothing
    prefetched).  Rows are independent, so the two must agree bit for bit - a context image that lands early or late, or
    a workgroup that prefetches past its last group, shows up here."""
    enc, cfg, sd = enc_l2
    B, S = 328, 256
    rng = np.random.default_rng(31)
    lengths = [int(x) for x in rng.integers(2, S + 1, size=B)]
    for b in range(0, B, 7):
        lengths[b] = S
    ids, mask = enc_oracle.synthetic_token_ids(B, S, seed=99, lengths=lengths)
    whole = enc.encode_token_ids(ids, mask).cpu().numpy()
    assert np.isfinite(whole).all()
    for lo in range(0, B, 64):
        part = enc.encode_token_ids(ids[lo:lo + 64], mask[lo:lo + 64]).cpu().numpy()
        assert np.array_equal(part, whole[lo:lo + 64]), f"rows {lo}..{lo + 63}"


def test_padding_invariance(enc_l2):
    """An embedding must not depend on batch-mates or on how far the row is padded."""
    enc, cfg, sd = enc_l2
    ids, mask = enc_oracle.synthetic_token_ids(4, 96, seed=9, lengths=[96, 40, 23, 64])
    full = enc.encode_token_ids(ids, mask).cpu().numpy()
    alone = enc.encode_token_ids(ids[1:2, :40], mask[1:2, :40]).cpu().numpy()
    np.testing.assert_allclose(alone[0], full[1], atol=1e-5)
    # pad token ids behind the mask are irrelevant
    ids2 = ids.copy()
    ids2[1, 40:] = 777
    again = enc.encode_token_ids(ids2, mask).cpu().numpy()
    assert np.array_equal(again[1], full[1])
    # determinism (test_model_validation.py:100-110 asks 1e-5; the kernels are bit-deterministic)
    assert np.array_equal(enc.encode_token_ids(ids, mask).cpu().numpy(), full)


def test_unnormalized_and_out_of_range_ids(enc_l2):
    enc, cfg, sd = enc_l2
    ids, mask = enc_oracle.synthetic_token_ids(2, 20, seed=3)
    raw = enc.encode_token_ids(ids, mask, normalize=False).cpu().numpy()
    want = enc_oracle.encode_token_ids(sd, ids, mask, cfg.num_hidden_layers, normalize=False)
    assert _cos(raw, want).min() >= COS_MIN
    np.testing.assert_allclose(np.linalg.norm(raw, axis=1), np.linalg.norm(want, axis=1), rtol=2e-2)
    with pytest.raises(_native.NativeError, match="exceeds max"):
        enc.encode_token_ids(np.zeros((1, 513), np.int32))


def _vocab():
    words = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    words += ["query", "passage", ":", "what", "is", "machine", "learning", "how", "does", "search", "work",
              "semantic", "the", "a", "of", "deep", "neural", "network", "##s", "##ing", "vector", "index",
              "hello", "world", "test", "document", "text", "##1", "##2", "##3", ".", "?", "!", "e", "##5"]
    return words


def test_text_path_and_student_model(gpu, tmp_path):
    """End to end: local model directory (config.json + model.safetensors + vocab.txt) ->
    StudentModel -> encode / encode_queries / encode_documents, compared with the oracle run on the
    very ids the tokenizer produced (text -> ids parity is unpinned offline: no e5 vocab on disk)."""
    vocab = _vocab()
    cfg = BertConfig(vocab_size=len(vocab), num_hidden_layers=2)
    sd = synthetic_state_dict(cfg)
    mdir = tmp_path / "e5-small-v2-synthetic"
    save_model_dir(mdir, cfg, sd)
    (mdir / "vocab.txt").write_text("\n".join(vocab))
    student = StudentModel(str(mdir), device="cuda:0")
    assert student.embedding_dim == 384 and student.device == "cuda:0" and student.is_e5
    texts = ["What is machine learning?", "hello world", "Semantic search networks!", ""]
    emb = student.encode(texts, batch_size=2)
    assert emb.shape == (4, 384) and emb.dtype == np.float32
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    tok = student.model.tokenize(texts)
    assert tok["input_ids"][1, 0] == 101 and tok["input_ids"][3].tolist()[:2] == [101, 102]  # empty text -> [CLS][SEP]
    want = enc_oracle.encode_token_ids(sd, tok["input_ids"], tok["attention_mask"], 2)
    assert _cos(emb, want).min() >= COS_MIN
    # single string in -> [1, 384] out through StudentModel.encode (wrapped into a list)
    one = student.encode("hello world")
    assert one.shape == (1, 384)
    # batch-mate independence: the text path packs sequences into shared rows, so only the grouping of
    # keys into 32-key tiles differs between the two calls (bf16 tolerance of SURVEY.md section 8f: 1e-2)
    np.testing.assert_allclose(one[0], emb[1], atol=2e-3)
    # E5 prefixes (tests/test_student_model.py:72-102)
    q = student.encode_queries("hello world")
    q_manual = student.encode(["query: hello world"])
    d = student.encode_documents(["hello world"], batch_size=4)
    d_manual = student.encode(["passage: hello world"])
    assert np.array_equal(q, q_manual) and np.array_equal(d, d_manual)
    assert not np.allclose(q, one)
    sims = student.compute_similarity(q, np.concatenate([d, emb]))
    assert sims.shape == (1, 5) and np.all(sims <= 1.0001) and np.all(sims >= -1.0001)
    np.testing.assert_allclose(sims, q @ np.concatenate([d, emb]).T, atol=1e-6)
    # long text is truncated to max_length and keeps [SEP]
    long = student.model.tokenize(["test " * 2000])
    assert long["input_ids"].shape[1] == student.max_length and long["input_ids"][0, -1] == 102
    student.cleanup()


def test_added_special_tokens_in_raw_text_follow_the_library(gpu):
    """A tokenizer.json that registers its special tokens as ADDED tokens (every real BERT one does) matches "[SEP]" /
    "[MASK]" in raw text as ONE id; the C++ WordPiece would split them.  Such texts are routed to the library: the flat
    id stream of the product's tokenisation equals the library's, text by text, with plain texts still on the C++ path."""
    from semantic_search_kd_amd.bench_support import synthetic_passages, synthetic_vocab
    from semantic_search_kd_amd.encoder import Mi355xSentenceEncoder, build_wordpiece_tokenizer

    vocab = synthetic_vocab(4000)
    tok = build_wordpiece_tokenizer(vocab)
    tok.add_special_tokens(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"])
    cfg = BertConfig(vocab_size=len(vocab), num_hidden_layers=1)
    enc = Mi355xSentenceEncoder(None, "cuda:0", config=cfg, state_dict=synthetic_state_dict(cfg), tokenizer=tok)
    assert enc._native_tok is not None and "[SEP]" in enc._native_tok.added_tokens
    docs = synthetic_passages(vocab, 30, seed=2)
    texts = docs[:10] + [docs[10] + " [SEP] " + docs[11], "[MASK] " + docs[12], "a [ sep ] b [brackets]", "[CLS]"] + docs[13:]
    flat, lengths = enc._tokenize_flat(texts)
    cu = np.concatenate([[0], np.cumsum(lengths)])
    for i, e in enumerate(tok.encode_batch(texts)):
        assert flat[cu[i] : cu[i + 1]].tolist() == e.ids, texts[i]
    sep = vocab.index("[SEP]")
    assert flat[cu[10] : cu[11]].tolist().count(sep) == 2     # the literal one and the template's


def test_model_name_is_never_fetched(gpu):
    with pytest.raises(FileNotFoundError, match="never downloads"):
        StudentModel("intfloat/e5-small-v2", device="cuda:0")
    with pytest.raises(RuntimeError, match="MI355X only"):
        StudentModel("whatever", device="cpu")


@pytest.mark.gpu
def test_captured_forward_replays_the_eager_forward(gpu):
    """Mi355xSentenceEncoder.capture_forward: the fixed-shape forward as one HIP graph - same bits as the eager
    launches, follows refilled inputs, refuses to run on weights that changed after the capture."""
    from semantic_search_kd_amd import BertConfig, Mi355xSentenceEncoder
    from semantic_search_kd_amd.bench_support import synthetic_ids

    cfg = BertConfig(num_hidden_layers=2)
    enc = Mi355xSentenceEncoder.from_synthetic(cfg, device="cuda:0")
    ids, mask = synthetic_ids(64, 96, cfg.vocab_size, torch.device("cuda:0"), seed=3)
    want = enc.encode_token_ids(ids, mask).clone()
    fwd = enc.capture_forward(ids, mask)
    assert torch.equal(fwd.replay(), want)
    ids2, mask2 = synthetic_ids(64, 96, cfg.vocab_size, torch.device("cuda:0"), seed=4)
    want2 = enc.encode_token_ids(ids2, mask2).clone()
    ids.copy_(ids2)
    mask.copy_(mask2)
    assert torch.equal(fwd.replay(), want2) and not torch.equal(want, want2)
    from semantic_search_kd_amd.weights import DeviceWeights

    enc.weights = DeviceWeights(cfg, enc._host_state, enc.device)      # e.g. after a training step re-tiled them
    with pytest.raises(RuntimeError, match="weights changed"):
        fwd.replay()


@pytest.mark.gpu
def test_two_branch_forward_matches_slices_eager_and_captured(gpu):
    """sskd_encoder_forward runs a batch whose halves each hold >= 2 groups of 128 tokens per CU (512 x 256 on 256 CUs) as
    TWO halves, one on a side stream forked from / joined back into the caller's stream (csrc/common.h
    run_parts_on_streams).  Rows do not interact, so the whole batch must equal the same rows in slices of 64 (one stream,
    one group per workgroup) bit for bit - eagerly, and replayed as one captured HIP graph (the side stream joins the
    capture; refilled inputs are followed)."""
    from semantic_search_kd_amd.bench_support import synthetic_ids

    cfg = BertConfig(num_hidden_layers=2)
    enc = Mi355xSentenceEncoder.from_synthetic(cfg, device="cuda:0")
    dev = torch.device("cuda:0")
    B, S = 512, 256
    ids, mask = synthetic_ids(B, S, cfg.vocab_size, dev, seed=11)
    mask[5, 40:] = 0          # a few ragged rows in both halves
    mask[300, 7:] = 0
    mask[511, 200:] = 0

    def in_slices(i, m):
        return torch.cat([enc.encode_token_ids(i[lo:lo + 64], m[lo:lo + 64]).clone() for lo in range(0, B, 64)])

    want = in_slices(ids, mask)
    got = enc.encode_token_ids(ids, mask).clone()
    assert torch.isfinite(got).all()
    assert torch.equal(got, want)
    fwd = enc.capture_forward(ids, mask)
    assert torch.equal(fwd.replay(), want)
    ids2, mask2 = synthetic_ids(B, S, cfg.vocab_size, dev, seed=12)
    want2 = in_slices(ids2, mask2)
    ids.copy_(ids2)
    mask.copy_(mask2)
    assert torch.equal(fwd.replay(), want2) and not torch.equal(want, want2)
