"""Request / response models of the search API.

Field names, defaults and bounds follow the reference's wire format (src/serve/schemas.py:8-136):
``k`` in [1, 100], ``rerank_top_k`` in [1, 200], query length in [1, 1000], 1..100 texts per
``/encode`` call.
"""
from __future__ import annotations

from typing import List, Optional

from pydantic import BaseModel, Field


class SearchRequest(BaseModel):
    query: str = Field(..., min_length=1, max_length=1000, description="query text")
    k: int = Field(10, ge=1, le=100, description="results to return")
    rerank: bool = Field(False, description="re-score the candidates with the teacher cross-encoder")
    rerank_top_k: int = Field(50, ge=1, le=200, description="candidates retrieved when reranking")


class SearchResult(BaseModel):
    doc_id: str
    text: str
    score: float
    rank: int = Field(..., description="1-based position")


class SearchResponse(BaseModel):
    query: str
    results: List[SearchResult]
    total_results: int
    reranked: bool
    latency_ms: float


class EncodeRequest(BaseModel):
    texts: List[str] = Field(..., min_length=1, max_length=100)
    normalize: bool = True


class EncodeResponse(BaseModel):
    embeddings: List[List[float]]
    dimension: int
    num_texts: int
    latency_ms: float


class HealthResponse(BaseModel):
    status: str
    model_loaded: bool
    index_loaded: bool
    index_size: int
    version: str


class ErrorResponse(BaseModel):
    error: str
    detail: Optional[str] = None
