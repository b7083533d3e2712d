"""API contract of the serving surface (CPU, backends mocked like the reference's tests/test_api.py)."""
from unittest.mock import MagicMock

import numpy as np
import pytest
from fastapi.testclient import TestClient

from semantic_search_kd_amd.serve import app as app_module
from semantic_search_kd_amd.serve.app import ServeSettings, app_state, create_app


def _mock_student():
    m = MagicMock()
    m.embedding_dim = 384
    m.max_length = 512

    def enc(texts, **kw):
        rng = np.random.default_rng(len(str(texts)))
        e = rng.standard_normal((len(texts), 384)).astype(np.float32)
        return e / np.linalg.norm(e, axis=1, keepdims=True)

    m.encode.side_effect = enc
    m.encode_queries.side_effect = enc
    m.encode_documents.side_effect = enc
    return m


@pytest.fixture
def client():
    for k, v in vars(app_module.AppState()).items():
        setattr(app_state, k, v)
    app_state.student = _mock_student()
    app = create_app(settings=ServeSettings(environment="test"))
    with TestClient(app) as c:
        yield c
    for k, v in vars(app_module.AppState()).items():
        setattr(app_state, k, v)


def test_root_health_live_ready(client):
    r = client.get("/")
    assert r.status_code == 200 and r.json()["status"] == "running" and r.json()["service"] == "Semantic Search API"
    h = client.get("/health").json()
    assert set(h) == {"status", "model_loaded", "index_loaded", "index_size", "version"}     # test_api.py:200-215
    assert h["status"] == "healthy" and h["model_loaded"] and not h["index_loaded"] and h["index_size"] == 0
    assert client.get("/live").json() == {"alive": True}
    assert client.get("/ready").json() == {"ready": True}
    app_state.ready = False
    assert client.get("/ready").status_code == 503


def test_encode_route(client):
    r = client.post("/encode", json={"texts": ["hello", "world"], "normalize": True})
    assert r.status_code == 200
    body = r.json()
    assert body["num_texts"] == 2 and body["dimension"] == 384 and len(body["embeddings"][0]) == 384
    # plain encode (no E5 prefix) on this route — reference app.py:385-389
    app_state.student.encode.assert_called_once()
    assert app_state.student.encode.call_args[0][0] == ["hello", "world"]
    assert app_state.student.encode.call_args[1] == {"convert_to_numpy": True, "normalize": True}
    assert client.post("/encode", json={"texts": []}).status_code == 422                      # test_api.py:296-306
    assert client.post("/encode", json={"texts": ["x"] * 101}).status_code == 422             # schemas.py:69-71


def test_search_validation_and_unavailable(client):
    assert client.post("/search", json={"query": "q", "k": 5}).status_code == 503             # test_api.py:312-320
    assert client.post("/search", json={"query": "q", "k": 5}).json()["error"] == "Search index not loaded"
    assert client.post("/search", json={"k": 5}).status_code == 422                           # missing query
    assert client.post("/search", json={"query": "", "k": 5}).status_code == 422
    assert client.post("/search", json={"query": "q", "k": 0}).status_code == 422
    assert client.post("/search", json={"query": "q", "k": 101}).status_code == 422           # schemas.py:12
    assert client.post("/search", json={"query": "q" * 1001}).status_code == 422
    assert client.get("/nonexistent").status_code == 404
    assert client.get("/search").status_code == 405


def test_search_glue_semantics(client):
    """idx -> doc_id -> text mapping, -1 / out-of-range skipping with rank gaps, trim to k (app.py:285-352)."""
    index = MagicMock()
    index.search.return_value = (
        np.array([[0.9, 0.8, 0.7, 0.6, -3.4e38]], np.float32),
        np.array([[2, 7, 0, 1, -1]], np.int64),
    )
    app_state.index_builder = index
    app_state.doc_ids = ["a", "b", "c"]
    app_state.doc_texts = {"a": "text a", "c": "text c"}
    r = client.post("/search", json={"query": "what is x", "k": 5})
    assert r.status_code == 200
    body = r.json()
    app_state.student.encode_queries.assert_called_once_with(["what is x"])                   # app.py:287
    assert index.search.call_args[1] == {"k": 5}
    got = [(x["doc_id"], x["text"], x["rank"]) for x in body["results"]]
    assert got == [("c", "text c", 1), ("a", "text a", 3), ("b", "", 4)]                       # idx 7 and -1 skipped, ranks keep gaps
    assert body["total_results"] == 3 and body["reranked"] is False and body["latency_ms"] >= 0
    assert abs(body["results"][0]["score"] - 0.9) < 1e-6
    # rerank asked, no teacher loaded: retrieves rerank_top_k, returns unreranked, trimmed to k
    r = client.post("/search", json={"query": "q", "k": 2, "rerank": True, "rerank_top_k": 4})
    assert index.search.call_args[1] == {"k": 4}
    assert r.json()["reranked"] is False and len(r.json()["results"]) == 2
    # with a teacher: scores overwritten, re-sorted, ranks renumbered (app.py:321-339)
    teacher = MagicMock()
    teacher.score.return_value = [0.1, 5.0, 2.0]
    app_state.teacher = teacher
    r = client.post("/search", json={"query": "q", "k": 2, "rerank": True, "rerank_top_k": 4}).json()
    assert [(x["doc_id"], x["rank"]) for x in r["results"]] == [("a", 1), ("b", 2)] and r["reranked"] is True
    assert teacher.score.call_args[0][0][0] == ["q", "text c"]
    # backend failure -> 500 with the reference's message shape (app.py:356-361)
    index.search.side_effect = RuntimeError("HIP error")
    r = client.post("/search", json={"query": "q"})
    assert r.status_code == 500 and r.json()["error"] == "Search failed: HIP error"


def test_sharded_failure_is_a_500_and_health_reflects_shard_state(client):
    """reference: src/serve/app.py:354-361; SURVEY.md section 5 - a failed rank of a sharded index surfaces as the
    route's 500, and /health.index_loaded follows the shards' state, not the mere existence of rank 0's object."""
    from semantic_search_kd_amd.sharded_index import ShardedIndex, ShardFailure

    index = ShardedIndex()
    app_state.index_builder, app_state.doc_ids = index, ["a", "b"]
    assert client.get("/health").json()["index_loaded"] is False          # nothing loaded on the ranks yet
    index.local = object()
    assert client.get("/health").json()["index_loaded"] is True
    index.search = MagicMock(side_effect=ShardFailure("search", {3: "RuntimeError: HIP error: out of memory"}))
    r = client.post("/search", json={"query": "q"})
    assert r.status_code == 500 and "rank 3" in r.json()["error"] and r.json()["error"].startswith("Search failed:")
    index.broken = "search: a rank did not report within 60 s"
    assert client.get("/health").json()["index_loaded"] is False
    assert index.health()["loaded"] is False and index.health()["broken"]


def test_index_load_missing_dir_is_404(client):
    r = client.post("/index/load", params={"index_path": "/nonexistent/dir"})
    assert r.status_code == 404 and "Index not found" in r.json()["error"]


def test_request_response_contract_matches_the_reference_schemas():
    """tests/golden/api_schemas.json is the validation contract (field names, types, defaults, bounds,
    required fields) extracted from the reference's own src/serve/schemas.py by
    tests/golden/make_golden.py::make_api_schemas; the drop-in's models must state the same."""
    import json
    from pathlib import Path

    from semantic_search_kd_amd.serve import schemas as mine

    ref = json.loads((Path(__file__).resolve().parent / "golden" / "api_schemas.json").read_text())
    drop = {"title", "description", "example", "examples"}

    def strip(node):
        if isinstance(node, dict):
            return {k: strip(v) for k, v in sorted(node.items()) if k not in drop}
        if isinstance(node, list):
            return [strip(v) for v in node]
        return node

    assert set(ref) == {"SearchRequest", "SearchResult", "SearchResponse", "EncodeRequest", "EncodeResponse",
                        "HealthResponse", "ErrorResponse"}
    for name, contract in ref.items():
        assert strip(getattr(mine, name).model_json_schema()) == contract, name
