"""End-to-end serving on the GPU: a SUCCESSFUL search through the real backend — the test the
reference's suite never had (its temp_index_dir fixture, tests/conftest.py:175-200, is unused)."""
import json

import numpy as np
import pandas as pd
import pytest
from fastapi.testclient import TestClient

from oracle import encoder as enc_oracle
from oracle import search as oracle
from semantic_search_kd_amd import BertConfig, FAISSIndexBuilder, StudentModel, synthetic_state_dict
from semantic_search_kd_amd.build_index_cli import main as build_index_main
from semantic_search_kd_amd.serve import app as app_module
from semantic_search_kd_amd.serve.app import ServeSettings, app_state, create_app
from semantic_search_kd_amd.weights import save_model_dir
from test_encoder_gpu import _vocab

pytestmark = pytest.mark.gpu

DOCS = [
    "machine learning is a search of the vector index",
    "deep neural networks work",
    "hello world test document",
    "what is semantic search?",
    "the index of a document text",
    "how does a neural network work?",
]


@pytest.fixture
def model_dir(tmp_path):
    vocab = _vocab()
    cfg = BertConfig(vocab_size=len(vocab), num_hidden_layers=2)
    sd = synthetic_state_dict(cfg)
    mdir = tmp_path / "e5-small-v2-synthetic"
    save_model_dir(mdir, cfg, sd)
    (mdir / "vocab.txt").write_text("\n".join(vocab))
    return mdir, sd


def test_build_index_cli_then_serve(gpu, tmp_path, model_dir):
    mdir, sd = model_dir
    corpus = tmp_path / "corpus.parquet"
    pd.DataFrame(
        {"chunk_id": [f"chunk_{i}" for i in range(len(DOCS))], "text": DOCS, "doc_id": [f"doc_{i}" for i in range(len(DOCS))]}
    ).to_parquet(corpus)  # reference corpus schema: tests/conftest.py:210-216
    out = tmp_path / "index"
    rc = build_index_main([
        "--model-path", str(mdir), "--data-path", str(corpus), "--output-dir", str(out),
        "--batch-size", "4", "--device", "cuda:0", "--hnsw-m", "32", "--hnsw-ef-construction", "200",
    ])
    assert rc == 0
    assert json.loads((out / "doc_ids.json").read_text()) == [f"chunk_{i}" for i in range(len(DOCS))]
    assert json.loads((out / "texts.json").read_text())["chunk_2"] == DOCS[2]

    for k, v in vars(app_module.AppState()).items():
        setattr(app_state, k, v)
    app = create_app(student_model_path=str(mdir), device="cuda:0", settings=ServeSettings(environment="test"))
    with TestClient(app) as client:
        assert client.get("/health").json()["model_loaded"] is True
        r = client.post("/index/load", params={"index_path": str(out)})
        assert r.status_code == 200 and r.json() == {"status": "loaded", "index_path": str(out), "num_documents": 6}
        assert client.get("/health").json()["index_size"] == 6
        q = "what is semantic search?"
        r = client.post("/search", json={"query": q, "k": 3})
        assert r.status_code == 200
        body = r.json()
        assert body["total_results"] == 3 and [x["rank"] for x in body["results"]] == [1, 2, 3]
        # expected ranking from the oracle on the ids the tokenizer produced
        student = app_state.student
        dt = student.model.tokenize(["passage: " + d for d in DOCS])
        qt = student.model.tokenize(["query: " + q])
        de = enc_oracle.encode_token_ids(sd, dt["input_ids"], dt["attention_mask"], 2)
        qe = enc_oracle.encode_token_ids(sd, qt["input_ids"], qt["attention_mask"], 2)
        ref_s, ref_i = oracle.topk_blas(qe, de, 3)
        assert [x["doc_id"] for x in body["results"]] == [f"chunk_{i}" for i in ref_i[0]]
        assert body["results"][0]["text"] == DOCS[ref_i[0, 0]]
        np.testing.assert_allclose([x["score"] for x in body["results"]], ref_s[0], atol=5e-3)
        # k larger than the corpus: all 6 documents, no padding entries leak out
        r = client.post("/search", json={"query": q, "k": 50}).json()
        assert r["total_results"] == 6
        # /encode is unit-norm and unprefixed
        e = np.array(client.post("/encode", json={"texts": ["hello world"]}).json()["embeddings"])
        np.testing.assert_allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-5)
    for k, v in vars(app_module.AppState()).items():
        setattr(app_state, k, v)


def test_build_from_parquet_max_docs(gpu, tmp_path, model_dir):
    mdir, _ = model_dir
    corpus = tmp_path / "c.parquet"
    pd.DataFrame({"chunk_id": [f"c{i}" for i in range(len(DOCS))], "text": DOCS}).to_parquet(corpus)
    student = StudentModel(str(mdir), device="cuda:0")
    b = FAISSIndexBuilder(embedding_dim=384, index_type="HNSW", metric="cosine")
    index = b.build_from_parquet(model=student, parquet_path=corpus, batch_size=2, max_docs=4)
    assert index.ntotal == 4 and b.doc_ids == ["c0", "c1", "c2", "c3"]
    D, I = b.search(student.encode_queries(["hello world"]), k=10)
    assert (I[0, :4] >= 0).all() and (I[0, 4:] == -1).all()
