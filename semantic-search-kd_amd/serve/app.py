"""FastAPI application serving ``/search`` and ``/encode`` from the MI355X backend.

Route set, status codes and response shapes mirror the reference (src/serve/app.py:221-457):
``/``, ``/health``, ``/ready``, ``/live``, ``POST /search``, ``POST /encode``, ``POST /index/load``.
Glue semantics kept on purpose (src/serve/app.py:285-352): the query is encoded with
``encode_queries`` (E5 prefix), ``/encode`` uses plain ``encode``; ids < 0 or past ``doc_ids`` are
skipped *without* renumbering ranks; ``score = float(distance)``; results are trimmed to ``k``
after the optional rerank; any non-HTTP exception becomes a 500 ``{"error": "Search failed: ..."}``.
The reference's auth / rate-limit / logging middlewares (src/serve/middleware.py) are HTTP hygiene
outside the compute path and are not re-implemented here; they can be mounted in front unchanged.
"""
from __future__ import annotations

import json
import logging
import os
import time
from contextlib import asynccontextmanager
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional

from fastapi import FastAPI, HTTPException, Request, status
from fastapi.responses import JSONResponse

logger = logging.getLogger("semantic_search_kd_amd.serve")

from ..index import FAISSIndexBuilder  # noqa: E402,F401
from ..sharded_index import open_index
from ..student import StudentModel
from .schemas import (
    EncodeRequest,
    EncodeResponse,
    ErrorResponse,
    HealthResponse,
    SearchRequest,
    SearchResponse,
    SearchResult,
)

VERSION = "1.1.0"  # API version of the reference this surface is wire-compatible with (app.py:41)


@dataclass
class ServeSettings:
    """The few settings the path reads (reference: src/config.py:22-32,223-233; env prefix
    ``SEMANTIC_KD_`` with ``__`` nesting, src/config.py:275-279)."""

    student_model_name: str = "./artifacts/models/kd_student_production"
    student_device: Optional[str] = "cuda"
    index_dir: Optional[str] = None
    environment: str = "development"
    extra: Dict[str, Any] = field(default_factory=dict)

    @staticmethod
    def from_env() -> "ServeSettings":
        e = os.environ
        return ServeSettings(
            student_model_name=e.get("SEMANTIC_KD_STUDENT__MODEL_NAME", ServeSettings.student_model_name),
            student_device=e.get("SEMANTIC_KD_STUDENT__DEVICE", "cuda"),
            index_dir=e.get("SEMANTIC_KD_INDEX__DIR"),
            environment=e.get("SEMANTIC_KD_ENVIRONMENT", "development"),
        )

    def is_production(self) -> bool:
        return self.environment == "production"


class AppState:
    """Module-global holder of the duck-typed backend objects (reference: app.py:49-66)."""

    def __init__(self) -> None:
        self.student = None
        self.teacher = None
        self.index_builder = None
        self.doc_ids: Optional[List[str]] = None
        self.doc_texts: Optional[Dict[str, str]] = None
        self.settings: Optional[ServeSettings] = None
        self.ready: bool = False

    def is_ready(self) -> bool:
        return self.ready and self.student is not None


app_state = AppState()


def _load_index_dir(index_dir: Path) -> Dict[str, Any]:
    # a directory with a shards.json manifest opens as a row-sharded index (sharded_index.ShardedIndex: same
    # search / doc_ids surface, answered by every rank of the process group); anything else as one FAISSIndexBuilder
    builder = open_index(index_dir, app_state.student.embedding_dim, current=app_state.index_builder)
    app_state.index_builder = builder
    app_state.doc_ids = builder.doc_ids
    texts_path = index_dir / "texts.json"
    if texts_path.exists():
        with open(texts_path) as f:
            app_state.doc_texts = json.load(f)
    elif builder.doc_texts:
        app_state.doc_texts = builder.doc_texts
    return {"status": "loaded", "index_path": str(index_dir), "num_documents": len(app_state.doc_ids)}


def create_app(
    student_model_path: Optional[str] = None,
    teacher_model_path: Optional[str] = None,
    index_dir: Optional[Path] = None,
    device: str = "cuda",
    settings: Optional[ServeSettings] = None,
) -> FastAPI:
    """Application factory with the reference's signature (app.py:124-130)."""
    settings = settings or ServeSettings.from_env()
    if student_model_path:
        settings.student_model_name = student_model_path
    if device:
        settings.student_device = device
    if index_dir:
        settings.index_dir = str(index_dir)

    @asynccontextmanager
    async def lifespan(app: FastAPI):
        app_state.settings = settings
        if app_state.student is None:
            app_state.student = StudentModel(model_name=settings.student_model_name, device=settings.student_device)
        if teacher_model_path and app_state.teacher is None:
            # reference: app.py:96-107 - a teacher that fails to load only disables reranking
            try:
                from ..teacher import TeacherModel

                app_state.teacher = TeacherModel(model_name=teacher_model_path, device=settings.student_device)
            except Exception as exc:  # noqa: BLE001
                logger.warning("Failed to load teacher model (reranking disabled): %s", exc)
        if settings.index_dir and app_state.index_builder is None and Path(settings.index_dir).exists():
            _load_index_dir(Path(settings.index_dir))
        app_state.ready = True
        yield
        app_state.ready = False

    app = FastAPI(title="Semantic Search API", version=VERSION, lifespan=lifespan)
    register_routes(app, settings)
    return app


def register_routes(app: FastAPI, settings: ServeSettings) -> None:
    @app.get("/", response_model=Dict[str, Any])
    async def root() -> Dict[str, Any]:
        return {
            "service": "Semantic Search API",
            "version": VERSION,
            "status": "running" if app_state.is_ready() else "starting",
            "environment": settings.environment,
        }

    @app.get("/health", response_model=HealthResponse)
    async def health() -> HealthResponse:
        return HealthResponse(
            status="healthy" if app_state.is_ready() else "unhealthy",
            model_loaded=app_state.student is not None,
            # a row-sharded index counts as loaded only while EVERY rank holds its shards and answers
            # (sharded_index.ShardedIndex.is_loaded)
            index_loaded=app_state.index_builder is not None and bool(getattr(app_state.index_builder, "is_loaded", True)),
            index_size=len(app_state.doc_ids) if app_state.doc_ids else 0,
            version=VERSION,
        )

    @app.get("/ready")
    async def readiness() -> Dict[str, bool]:
        if not app_state.is_ready():
            raise HTTPException(status_code=status.HTTP_503_SERVICE_UNAVAILABLE, detail="Service not ready")
        return {"ready": True}

    @app.get("/live")
    async def liveness() -> Dict[str, bool]:
        return {"alive": True}

    @app.post("/search", response_model=SearchResponse)
    async def search(request: SearchRequest) -> SearchResponse:
        t0 = time.time()
        if app_state.student is None:
            raise HTTPException(status_code=status.HTTP_503_SERVICE_UNAVAILABLE, detail="Student model not loaded")
        if app_state.index_builder is None:
            raise HTTPException(status_code=status.HTTP_503_SERVICE_UNAVAILABLE, detail="Search index not loaded")
        try:
            query_emb = app_state.student.encode_queries([request.query])
            k_retrieve = request.rerank_top_k if request.rerank else request.k
            distances, indices = app_state.index_builder.search(query_emb, k=k_retrieve)
            results: List[SearchResult] = []
            for rank, (dist, idx) in enumerate(zip(distances[0], indices[0]), 1):
                if idx < 0 or (app_state.doc_ids and idx >= len(app_state.doc_ids)):
                    continue
                doc_id = app_state.doc_ids[idx] if app_state.doc_ids else f"doc_{idx}"
                text = app_state.doc_texts.get(doc_id, "") if app_state.doc_texts else ""
                results.append(SearchResult(doc_id=doc_id, text=text, score=float(dist), rank=rank))
            reranked = False
            if request.rerank and app_state.teacher is not None and results:
                scores = app_state.teacher.score([[request.query, r.text] for r in results])
                for r, s in zip(results, scores):
                    r.score = float(s)
                results = sorted(results, key=lambda r: r.score, reverse=True)
                for pos, r in enumerate(results, 1):
                    r.rank = pos
                reranked = True
            results = results[: request.k]
            return SearchResponse(
                query=request.query,
                results=results,
                total_results=len(results),
                reranked=reranked,
                latency_ms=(time.time() - t0) * 1000,
            )
        except HTTPException:
            raise
        except Exception as exc:  # noqa: BLE001 - mirrors the reference's catch-all (app.py:356-361)
            raise HTTPException(
                status_code=status.HTTP_500_INTERNAL_SERVER_ERROR, detail=f"Search failed: {exc}"
            ) from exc

    @app.post("/encode", response_model=EncodeResponse)
    async def encode(request: EncodeRequest) -> EncodeResponse:
        t0 = time.time()
        if app_state.student is None:
            raise HTTPException(status_code=status.HTTP_503_SERVICE_UNAVAILABLE, detail="Model not loaded")
        try:
            emb = app_state.student.encode(request.texts, convert_to_numpy=True, normalize=request.normalize)
            return EncodeResponse(
                embeddings=emb.tolist(),
                dimension=emb.shape[1],
                num_texts=len(request.texts),
                latency_ms=(time.time() - t0) * 1000,
            )
        except Exception as exc:  # noqa: BLE001
            raise HTTPException(
                status_code=status.HTTP_500_INTERNAL_SERVER_ERROR, detail=f"Encoding failed: {exc}"
            ) from exc

    @app.post("/index/load")
    async def load_index(index_path: str) -> Dict[str, Any]:
        try:
            index_dir = Path(index_path)
            if not index_dir.exists():
                raise HTTPException(status_code=status.HTTP_404_NOT_FOUND, detail=f"Index not found: {index_path}")
            return _load_index_dir(index_dir)
        except HTTPException:
            raise
        except Exception as exc:  # noqa: BLE001
            raise HTTPException(
                status_code=status.HTTP_500_INTERNAL_SERVER_ERROR, detail=f"Failed to load index: {exc}"
            ) from exc

    @app.exception_handler(HTTPException)
    async def http_exception_handler(request: Request, exc: HTTPException) -> JSONResponse:
        return JSONResponse(status_code=exc.status_code, content=ErrorResponse(error=str(exc.detail)).model_dump())

    @app.exception_handler(Exception)
    async def general_exception_handler(request: Request, exc: Exception) -> JSONResponse:
        prod = app_state.settings.is_production() if app_state.settings else False
        return JSONResponse(
            status_code=status.HTTP_500_INTERNAL_SERVER_ERROR,
            content=ErrorResponse(error="Internal server error", detail=None if prod else str(exc)).model_dump(),
        )
