"""Host-side logic that needs no GPU: weight recipe, fragment tiling, index file format,
StudentModel behaviour (encoder mocked, like the reference's tests/test_student_model.py)."""
import json
import struct
from unittest.mock import MagicMock, patch

import numpy as np
import pytest

from semantic_search_kd_amd import BertConfig, read_flat_ip, synthetic_state_dict, write_flat_ip
from semantic_search_kd_amd import weights as W


# ------------------------------------------------------------------ weights
def test_synthetic_weights_are_platform_independent():
    t = W.synthetic_tensor("embeddings.word_embeddings.weight", (4,), 1.0)
    # splitmix64 known-answer: first output for seed 0 is 0xE220A8397B1DCDAF
    assert int(W._splitmix64(np.array([0], np.uint64))[0]) == 0xE220A8397B1DCDAF
    assert t.dtype == np.float32 and np.all(np.abs(t) < 1.0)
    again = W.synthetic_tensor("embeddings.word_embeddings.weight", (4,), 1.0)
    other = W.synthetic_tensor("embeddings.position_embeddings.weight", (4,), 1.0)
    assert np.array_equal(t, again) and not np.array_equal(t, other)


def test_state_dict_has_e5_small_parameter_count():
    """33 212 160 parameters without the (unused) pooler — SURVEY.md App. B."""
    sd = synthetic_state_dict(BertConfig())
    assert sum(v.size for v in sd.values()) == 33_212_160
    assert sd["encoder.layer.11.intermediate.dense.weight"].shape == (1536, 384)
    assert abs(float(np.std(sd["encoder.layer.0.output.dense.weight"])) - 0.02) < 1e-3


def test_weight_fragment_tiling_matches_kernel_contract():
    w = np.arange(64 * 48, dtype=np.float32).reshape(64, 48)
    t = W.tile_weight_fragments(w)
    assert t.shape == (2, 3, 64, 8)
    for nt, s, lane, j in ((0, 0, 0, 0), (1, 2, 63, 7), (0, 1, 37, 3), (1, 0, 5, 6)):
        r, h = lane & 31, lane >> 5
        assert t[nt, s, lane, j] == w[32 * nt + r, 16 * s + 8 * h + j]


def test_bf16_rounding_is_nearest_even():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.1415927], np.float32)
    bits = W.f32_to_bf16_bits(x)
    assert bits[0] == 0x3F80
    assert bits[1] == 0x3F80        # 1 + 2^-8 is a tie -> even (1.0)
    assert bits[2] == 0x3F82        # 1 + 3*2^-8 is a tie -> even mantissa
    assert np.allclose(W.bf16_round(x), x, rtol=2 ** -8)


def test_model_dir_roundtrip(tmp_path):
    cfg = BertConfig(vocab_size=300, num_hidden_layers=1)
    sd = synthetic_state_dict(cfg)
    W.save_model_dir(tmp_path / "m", cfg, sd)
    assert W.load_config(tmp_path / "m") == cfg
    back = W.load_state_dict(tmp_path / "m")
    assert set(back) == set(sd) and all(np.array_equal(back[k], sd[k]) for k in sd)
    with pytest.raises(FileNotFoundError, match="LOCAL model directory"):
        W.load_state_dict(tmp_path / "missing")


# ------------------------------------------------------------------ index file
def test_flat_index_file_layout(tmp_path):
    """index.faiss is written in faiss' IndexFlatIP layout (fourcc IxFI) — tests/conftest.py:184-188."""
    v = np.random.default_rng(0).standard_normal((7, 384)).astype(np.float32)
    write_flat_ip(tmp_path / "index.faiss", v)
    raw = (tmp_path / "index.faiss").read_bytes()
    assert raw[:4] == b"IxFI"
    d, n, _, _, trained, metric = struct.unpack("<iqqqBi", raw[4:4 + 33])
    assert (d, n, trained, metric) == (384, 7, 1, 0)
    assert struct.unpack("<Q", raw[37:45])[0] == 7 * 384 and len(raw) == 45 + 7 * 384 * 4
    assert np.array_equal(np.asarray(read_flat_ip(tmp_path / "index.faiss")), v)
    (tmp_path / "hnsw.faiss").write_bytes(b"IHNf" + b"\0" * 64)
    with pytest.raises(ValueError, match="HNSW graph files"):
        read_flat_ip(tmp_path / "hnsw.faiss")
    write_flat_ip(tmp_path / "empty.faiss", np.zeros((0, 384), np.float32))
    assert read_flat_ip(tmp_path / "empty.faiss").shape == (0, 384)


def test_builder_argument_validation():
    from semantic_search_kd_amd import FAISSIndexBuilder

    with pytest.raises(ValueError, match="384"):
        FAISSIndexBuilder(embedding_dim=768)
    with pytest.raises(ValueError, match="metric"):
        FAISSIndexBuilder(embedding_dim=384, metric="l2")


# ------------------------------------------------------------------ StudentModel
def _mock_encoder(n=1):
    m = MagicMock()
    m.get_sentence_embedding_dimension.return_value = 384
    m.max_seq_length = 512
    m.encode.return_value = np.random.randn(n, 384).astype(np.float32)
    return m


@patch("semantic_search_kd_amd.student.SentenceTransformer")
def test_default_device_falls_back_to_cpu_string(mock_st):
    """reference tests/test_student_model.py:12-24 (device=None -> "cpu" without a GPU)."""
    mock_st.return_value = _mock_encoder()
    with patch("semantic_search_kd_amd.student.torch") as mock_torch:
        mock_torch.cuda.is_available.return_value = False
        from semantic_search_kd_amd.student import StudentModel

        model = StudentModel(model_name="test-model", device=None)
        assert model.device == "cpu"
        mock_torch.cuda.is_available.return_value = True
        assert StudentModel(model_name="test-model").device == "cuda"
    mock_st.assert_called_with("test-model", device="cuda")


@patch("semantic_search_kd_amd.student.SentenceTransformer")
def test_encode_wraps_single_string_and_calls_encode_once(mock_st):
    """reference tests/test_student_model.py:38-70."""
    enc = _mock_encoder(1)
    mock_st.return_value = enc
    from semantic_search_kd_amd.student import StudentModel

    model = StudentModel(model_name="test-model", device="cpu")
    assert model.device == "cpu" and model.embedding_dim == 384 and model.max_length == 512
    model.encode("hello world")
    enc.encode.assert_called_once()
    args, kwargs = enc.encode.call_args
    assert args[0] == ["hello world"]
    assert kwargs["convert_to_numpy"] is True and kwargs["normalize_embeddings"] is True
    enc.encode.return_value = np.random.randn(3, 384).astype(np.float32)
    assert model.encode(["a", "b", "c"], batch_size=2, show_progress=True).shape == (3, 384)
    assert enc.encode.call_args[1]["batch_size"] == 2 and enc.encode.call_args[1]["show_progress_bar"] is True


@patch("semantic_search_kd_amd.student.SentenceTransformer")
def test_e5_prefixes(mock_st):
    """reference tests/test_student_model.py:72-102."""
    enc = _mock_encoder(1)
    mock_st.return_value = enc
    from semantic_search_kd_amd.student import StudentModel

    model = StudentModel(model_name="intfloat/e5-small-v2", device="cpu")
    model.encode_queries("test query")
    assert enc.encode.call_args[0][0] == ["query: test query"]
    model.encode_documents("test document")
    assert enc.encode.call_args[0][0] == ["passage: test document"]
    model.encode_documents(["a", "b"], batch_size=16, show_progress=True)
    assert enc.encode.call_args[0][0] == ["passage: a", "passage: b"] and enc.encode.call_args[1]["batch_size"] == 16
    plain = StudentModel(model_name="./artifacts/models/kd_student_production", device="cpu")
    plain.encode_queries(["q"])
    assert enc.encode.call_args[0][0] == ["q"]                      # no "e5" in the name -> no prefix (auto)
    forced = StudentModel(model_name="./artifacts/models/kd_student_production", device="cpu", prefix_mode="e5")
    forced.encode_queries(["q"])
    assert enc.encode.call_args[0][0] == ["query: q"]


@patch("semantic_search_kd_amd.student.SentenceTransformer")
def test_cleanup_is_safe_and_training_entry_delegates_to_the_encoder(mock_st):
    """reference tests/test_student_model.py:126-137, tests/test_hardening.py:432-453; the training entry
    (src/kd/train.py:180-187) tokenises and hands ids / mask to the encoder's trainable module."""
    enc = _mock_encoder()
    mock_st.return_value = enc
    from semantic_search_kd_amd.student import StudentModel

    model = StudentModel(model_name="test-model", device="cpu")
    model.cleanup()
    enc.tokenize.return_value = {"input_ids": "IDS", "attention_mask": "MASK"}
    out = model.encode_with_gradients("x", normalize=False)
    enc.tokenize.assert_called_once_with(["x"])
    enc.trainable.return_value.assert_called_once_with("IDS", "MASK", normalize=False)
    assert out is enc.trainable.return_value.return_value


def test_student_without_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from semantic_search_kd_amd import StudentModel

    with pytest.raises(RuntimeError, match="no CPU path|MI355X"):
        StudentModel("some/dir", device=None)


def test_bench_self_launches_ranks_from_a_bare_shell():
    """``python bench.py --gpus 2`` with no RANK in the environment starts torch.distributed.run as
    a child before touching the GPU and returns its exit code (checked here with the CPU-only
    ``--launch-check`` leg: rendezvous at 127.0.0.1 + one all-gather, rank 0 prints one JSON line)."""
    import json
    import os
    import subprocess
    import sys

    from conftest import REPO

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launch_check": True, "ranks": 2}
    # a failing child propagates its exit code
    bad = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--launch-check", "--steps", "x"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0


def test_native_wordpiece_equals_tokenizers_library():
    """The C++ host tokenizer (sskd_tokenizer_*) reproduces the `tokenizers` library - the engine
    behind the reference's SentenceTransformer.encode - id for id on ASCII text, and FLAGS every text
    with non-ASCII characters instead of approximating it."""
    import numpy as np

    from semantic_search_kd_amd.bench_support import synthetic_passages, synthetic_vocab
    from semantic_search_kd_amd.encoder import NativeWordPiece, build_wordpiece_tokenizer

    vocab = synthetic_vocab()
    tok = build_wordpiece_tokenizer(vocab)
    nt = NativeWordPiece.from_hf(tok)
    assert nt is not None
    texts = ["passage: " + d for d in synthetic_passages(vocab, 400, seed=9)]
    texts += ["", "   ", "Hello, World!! what's   up\t(with) THIS_thing?", "x" + "y" * 120 + " end",
              "bell\x07char and\x7fdel", "a.b-c $5.00 #tag @you [brackets] {braces} `tick` ~tilde^", "UPPER lower MiXeD",
              "tab\tnew\nline\rreturn", "query: how does semantic search work?", "123 4567 89.0", "[CLS] literal [SEP]"]
    rng = np.random.default_rng(0)
    alphabet = np.array(list("abcdefghijklmnopqrstuvwxyzABCXYZ0123456789     .,;:!?'\"()-_/\\@#$%&*+=<>[]{}|~^`\t\n"))
    texts += ["".join(rng.choice(alphabet, size=int(rng.integers(0, 300)))) for _ in range(300)]
    texts += ["café naïve", "日本語 text", "emoji \U0001F600 here"]
    for max_len in (512, 16):
        flat, lengths, uni = nt.encode_flat(texts, max_len)
        assert uni.tolist() == [any(ord(c) > 127 for c in t) for t in texts]
        cu = np.concatenate([[0], np.cumsum(lengths)])
        for i, e in enumerate(tok.encode_batch(texts)):
            if uni[i]:
                assert lengths[i] == 0
                continue
            want = e.ids if len(e.ids) <= max_len else e.ids[: max_len - 1] + e.ids[-1:]
            assert flat[cu[i] : cu[i + 1]].tolist() == want, texts[i]
    assert nt.encode_flat(["a\x00b", "c"], 512) is None  # NUL inside a text: caller uses the library path
    # a tokenizer that is not the uncased BERT recipe is never replaced
    from tokenizers import Tokenizer, models

    assert NativeWordPiece.from_hf(Tokenizer(models.WordPiece({"[UNK]": 0}, unk_token="[UNK]"))) is None
    # a real BERT tokenizer.json registers its special tokens as ADDED tokens: the library then matches "[SEP]" in raw
    # text as ONE id, the C++ path would split it - such texts are flagged for the library, everything else still agrees
    tok2 = build_wordpiece_tokenizer(vocab)
    tok2.add_special_tokens(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"])
    nt2 = NativeWordPiece.from_hf(tok2)
    assert nt2 is not None and {"[SEP]", "[MASK]", "[CLS]"} <= set(nt2.added_tokens)
    literal = "first part [SEP] second [MASK] part"
    assert nt2.needs_library(literal) and not nt2.needs_library("plain [brackets] and [ sep ] text")
    assert vocab.index("[SEP]") in tok2.encode(literal).ids[1:-1]           # the library: one id
    plain = texts[:50]
    flat, lengths, uni = nt2.encode_flat(plain, 512)
    cu = np.concatenate([[0], np.cumsum(lengths)])
    for i, e in enumerate(tok2.encode_batch(plain)):
        if not nt2.needs_library(plain[i]):
            assert flat[cu[i] : cu[i + 1]].tolist() == e.ids, plain[i]


def test_ance_miner_matches_reference_fixture():
    """tests/golden/ance_mining.json holds what the REFERENCE'S OWN ANCEMiner.mine
    (src/mining/miners.py:184-253) selected for a deterministic stand-in student
    (make_golden.make_ance); the batched drop-in must select the same ids in the same order."""
    import json
    import sys

    from conftest import GOLDEN

    sys.path.insert(0, str(GOLDEN))
    from make_golden import HashedEmbeddingStudent, ance_case

    from semantic_search_kd_amd.mining import ANCEMiner

    gold = json.loads((GOLDEN / "ance_mining.json").read_text())
    queries, positives, candidates, docs = ance_case()
    assert any(len(x) for x in gold["margin0.1_k5"]) and gold["margin0.3_k5"] != gold["margin0.0_k5"]
    for margin in (0.1, 0.3, 0.0):
        for top_k in (5, 2):
            got = ANCEMiner(HashedEmbeddingStudent(), margin=margin).mine(queries, positives, candidates, docs, docs, top_k=top_k)
            assert got == gold[f"margin{margin}_k{top_k}"], (margin, top_k)
    assert ANCEMiner(HashedEmbeddingStudent()).mine([], [], [], {}, {}) == []


def test_teacher_miner_matches_reference_fixture():
    """tests/golden/teacher_mining.json holds ids AND scores chosen by the REFERENCE'S OWN TeacherMiner.mine
    (src/mining/miners.py:104-158) for a deterministic stand-in teacher (make_golden.make_teacher_mining): the
    batched drop-in must return the same ids in the same order with the same scores - including the stable order
    of exact score ties, ids missing from the text table, empty candidate lists and the confidence filter - while
    calling ``teacher.score`` ONCE for all queries (the reference: once per query)."""
    import json
    import sys

    from conftest import GOLDEN

    sys.path.insert(0, str(GOLDEN))
    from make_golden import HashedTeacher, teacher_mining_case

    from semantic_search_kd_amd.mining import TeacherMiner

    gold = json.loads((GOLDEN / "teacher_mining.json").read_text())
    queries, candidates, docs = teacher_mining_case()
    assert gold["thr0.9_k10"] != gold["thr0.5_k10"] and any(len(x) < 3 for x in gold["thr0.9_k3"]["ids"])
    for thr in (0.6, 0.5, 0.9):
        for top_k in (10, 3):
            teacher = HashedTeacher()
            ids, scores = TeacherMiner(teacher, confidence_threshold=thr).mine(queries, candidates, docs, top_k=top_k)
            want = gold[f"thr{thr}_k{top_k}"]
            assert ids == want["ids"], (thr, top_k)
            assert scores == want["scores"], (thr, top_k)
            assert teacher.calls == [(sum(len(c) for c in candidates), 32)]
    assert TeacherMiner(HashedTeacher()).mine([], [], {}) == ([], [])
    assert TeacherMiner(HashedTeacher()).mine(["q"], [[]], {}) == ([[]], [[]])


def test_screened_search_launch_plan_is_sane_across_shapes():
    """The screening launch plan is a host function of the shape (no GPU needed): every served shape gets a
    positive workspace, 64 (small batches) or 128 / 160 queries per workgroup - whichever tiles the 256 CUs better -
    at least one slice and at least one tile per wave of a slice; unserved shapes are refused, not mis-planned."""
    import ctypes

    from semantic_search_kd_amd import _native

    lib = _native.load()
    for n, nq in [(2048, 64), (2049, 65), (4000, 10000), (125_000, 10_000), (1_000_000, 10_000), (8_841_823, 10_000),
                  (1_000_000, 1250), (1_000_000, 256), (1_000_000, 255), (33_000, 100_000), (70_001, 2999)]:
        qpb, passes, slices = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        rc = lib.sskd_index_search_screened_plan(n, nq, 10, ctypes.byref(qpb), ctypes.byref(passes), ctypes.byref(slices))
        assert rc == 0, (n, nq)
        assert qpb.value in ((128, 160) if nq >= 256 else (64,)), (n, nq, qpb.value)
        assert passes.value == -(-nq // qpb.value)
        tiles = -(-n // 32)
        assert 1 <= slices.value <= max(1, -(-tiles // 12)), (n, nq, slices.value)   # 12 waves per workgroup
        if (n, nq) in ((1_000_000, 10_000), (8_841_823, 10_000)):
            assert (qpb.value, slices.value) == (160, 4)        # 63 x 4 = 252 workgroups: one round of the chip
        if (n, nq) == (125_000, 10_000):
            assert (qpb.value, slices.value) == (128, 3)
        assert int(lib.sskd_index_search_screened_workspace_bytes(n, nq, 10)) > 0
        assert int(lib.sskd_index_bf16_bytes(n)) == tiles * 32 * 768 + 4096   # bf16 tiles + the norm block (no second fp32 copy since round 4)
    for n, nq, k in [(2047, 64, 10), (100_000, 63, 10), (100_000, 1000, 11), (100_000, 1000, 0)]:
        assert int(lib.sskd_index_search_screened_workspace_bytes(n, nq, k)) == 0
        assert lib.sskd_index_search_screened_plan(n, nq, k, None, None, None) != 0


def test_w2_image_is_permuted_for_the_lane_local_hand_over():
    """weights.tile_w2_chunked: slot j of lane l of fragment (chunk c, tile nt, k-step s2) must be
    W2[32 nt + (l & 31)][32 c + 16 s2 + 8 (j >> 2) + 4 (l >> 5) + (j & 3)] - the order in which a producer lane's
    accumulators (element 4g + e = hidden unit 8g + 4(l >> 5) + e) arrive, packed, as ITS OWN lane of the consumers' B
    fragments (csrc/encoder.hip fused_mlp_ln_kernel, "Hand-over").  The W1 image keeps the plain A-fragment order."""
    from semantic_search_kd_amd.weights import tile_w2_chunked, tile_weight_fragments

    w2 = np.arange(384 * 1536, dtype=np.float32).reshape(384, 1536)
    img = tile_w2_chunked(w2)
    assert img.shape == (48, 12, 2, 64, 8)
    c, nt, s2, l, j = np.meshgrid(np.arange(48), np.arange(12), np.arange(2), np.arange(64), np.arange(8), indexing="ij")
    want = w2[32 * nt + (l & 31), 32 * c + 16 * s2 + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3)]
    assert np.array_equal(img, want)
    # every weight appears exactly once
    assert np.array_equal(np.sort(img.ravel()), w2.ravel())
    w1 = np.arange(1536 * 384, dtype=np.float32).reshape(1536, 384)
    f = tile_weight_fragments(w1)
    nt, s, l, j = np.meshgrid(np.arange(48), np.arange(24), np.arange(64), np.arange(8), indexing="ij")
    assert np.array_equal(f, w1[32 * nt + (l & 31), 16 * s + 8 * (l >> 5) + j])
