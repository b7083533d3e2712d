"""Teacher (stage 2) and ANCE (stage 3) hard-negative mining on the MI355X teacher / encoder + exact scan.

``TeacherMiner`` is the drop-in for the reference's class of that name (src/mining/miners.py:80-158): same
constructor, same ``mine`` signature, same selection rule (stable descending sort by teacher score, the first
``top_k``, then the confidence filter) - but ONE ``teacher.score`` call over the pairs of every query instead of
one call of <= 100 pairs per query, so the GPU sees launches cut by a token budget, not by the query loop (and,
after ``TeacherModel.data_parallel()``, pair ranges sharded over the process group).


Drop-in for the reference's ``ANCEMiner`` (reference: src/mining/miners.py:160-253): same constructor,
same ``mine`` signature and the same selection rule -

    adversarial = candidates with  score >= max(positive scores) - margin   (0.0 when no positives)
    hard negatives = the ``top_k`` highest-scoring adversarial candidates, ties in candidate order

- but where the reference encodes one query, its positives and its candidates per loop iteration
(three ``encode`` calls per query), every text is encoded ONCE here, in a few large launches of the
packed varlen encoder, and the scores are ``q @ d.T`` on the GPU.  ``refresh`` + ``mine_from_index``
is the "ANCE refresh" use of the fast path (docs/adr-003: re-encode the corpus with the current
student every N steps, re-search it, mine): corpus -> ``FAISSIndexBuilder`` in HBM -> exact top-k.
The student is duck-typed exactly as in the reference (``encode_queries`` / ``encode_documents`` /
``compute_similarity``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np


def select_adversarial(cand_ids: Sequence[str], cand_scores: np.ndarray, pos_scores: np.ndarray,
                       margin: float, top_k: int) -> List[str]:
    """The reference's rule (src/mining/miners.py:232-247), including its stable descending sort."""
    max_pos = float(pos_scores.max()) if len(pos_scores) > 0 else 0.0
    adversarial = [(d, float(s)) for d, s in zip(cand_ids, cand_scores) if s >= max_pos - margin]
    adversarial.sort(key=lambda x: x[1], reverse=True)
    return [d for d, _ in adversarial[:top_k]]


def select_confident(cand_ids: Sequence[str], scores: Sequence[float], get_confidence, threshold: float, top_k: int):
    """The reference's rule (src/mining/miners.py:140-151): ``sorted(zip(ids, scores), key=score, reverse=True)`` is a
    STABLE descending sort (equal scores keep candidate order), the first ``top_k`` survive, then those whose
    confidence is below the threshold are dropped (so fewer than ``top_k`` may remain)."""
    ranked = sorted(zip(cand_ids, scores), key=lambda x: x[1], reverse=True)
    ids, kept = [], []
    for doc_id, score in ranked[:top_k]:
        if get_confidence(score) >= threshold:
            ids.append(doc_id)
            kept.append(score)
    return ids, kept


class TeacherMiner:
    def __init__(self, teacher_model, confidence_threshold: float = 0.6):
        self.teacher = teacher_model
        self.confidence_threshold = confidence_threshold

    def mine(self, queries: List[str], candidates: List[List[str]], candidate_texts: Dict[str, str], top_k: int = 10):
        """``(hard_negative_ids, teacher_scores)``, one list per query (src/mining/miners.py:104-158).  A candidate id
        missing from ``candidate_texts`` is scored against the empty text, as in the reference (:130)."""
        pairs = [(q, candidate_texts.get(doc_id, "")) for q, cand_ids in zip(queries, candidates) for doc_id in cand_ids]
        scores = list(self.teacher.score(pairs, batch_size=32)) if pairs else []
        all_ids: List[List[str]] = []
        all_scores: List[List[float]] = []
        at = 0
        for _, cand_ids in zip(queries, candidates):
            part = scores[at : at + len(cand_ids)]
            at += len(cand_ids)
            ids, kept = select_confident(cand_ids, part, self.teacher.get_confidence, self.confidence_threshold, top_k)
            all_ids.append(ids)
            all_scores.append(kept)
        return all_ids, all_scores


class ANCEMiner:
    def __init__(self, student_model, margin: float = 0.1):
        self.student = student_model
        self.margin = margin
        self._index = None
        self._corpus_ids: List[str] = []

    # ------------------------------------------------------------------ reference API
    def mine(
        self,
        queries: List[str],
        positives: List[List[str]],
        candidates: List[List[str]],
        candidate_texts: Dict[str, str],
        positive_texts: Dict[str, str],
        top_k: int = 5,
    ) -> List[List[str]]:
        if not queries:
            return []
        q_embs = np.asarray(self.student.encode_queries(list(queries)))
        # every distinct (role, doc id) text once; role matters because the two dicts may disagree
        slots: Dict[tuple, int] = {}
        texts: List[str] = []

        def slot(role: str, doc_id: str, table: Dict[str, str]) -> int:
            key = (role, doc_id)
            if key not in slots:
                slots[key] = len(texts)
                texts.append(table.get(doc_id, ""))
            return slots[key]

        pos_idx = [[slot("p", d, positive_texts) for d in ids] for ids in positives]
        cand_idx = [[slot("c", d, candidate_texts) for d in ids] for ids in candidates]
        d_embs = np.asarray(self.student.encode_documents(texts)) if texts else np.zeros((0, q_embs.shape[1]), np.float32)
        # ONE similarity launch per block of queries (every query x every distinct text; a (query, text) score does
        # not depend on its batch-mates), not two launches + two host round trips per query as in the reference loop
        out: List[List[str]] = []
        block = max(1, (1 << 24) // max(len(texts), 1))
        for q_lo in range(0, len(queries), block):
            sims = (np.asarray(self.student.compute_similarity(q_embs[q_lo : q_lo + block], d_embs))
                    if texts else np.zeros((len(q_embs[q_lo : q_lo + block]), 0), np.float32))
            for qi in range(q_lo, min(q_lo + block, len(queries))):
                row = sims[qi - q_lo]
                pos_scores = row[pos_idx[qi]] if pos_idx[qi] else np.zeros(0, np.float32)
                cand_scores = row[cand_idx[qi]] if cand_idx[qi] else np.zeros(0, np.float32)
                out.append(select_adversarial(candidates[qi], cand_scores, pos_scores, self.margin, top_k))
        return out

    # ------------------------------------------------------------------ ANCE refresh
    def refresh(self, corpus_ids: Sequence[str], corpus_texts: Sequence[str], device: Optional[str] = None):
        """Re-encode the corpus with the CURRENT student and rebuild the exact index in HBM."""
        from .index import FAISSIndexBuilder

        dev = device or getattr(self.student, "device", None)
        encode_device = getattr(self.student, "encode_documents_device", None)
        if encode_device is not None:
            embs = encode_device(list(corpus_texts))      # embeddings stay in HBM: encoder output -> index tiles
        else:
            embs = np.asarray(self.student.encode_documents(list(corpus_texts)))
        index = FAISSIndexBuilder(embedding_dim=int(embs.shape[1]), index_type="Flat", metric="ip", device=dev)
        index.add(embs)
        self._index, self._corpus_ids = index, list(corpus_ids)
        return index

    def mine_from_index(self, queries: List[str], positives: List[List[str]], top_k: int = 5,
                        search_k: int = 100) -> List[List[str]]:
        """Adversarial negatives straight from the refreshed index: the ``search_k`` nearest corpus
        rows of each query are its candidates (its positives excluded), then the same margin rule."""
        if self._index is None:
            raise RuntimeError("call refresh(corpus_ids, corpus_texts) first")
        if not queries:
            return []
        q_embs = np.ascontiguousarray(self.student.encode_queries(list(queries)), np.float32)
        scores, rows = self._index.search(q_embs, min(search_k, max(self._index.ntotal, 1)))
        row_of = {d: i for i, d in enumerate(self._corpus_ids)}
        out: List[List[str]] = []
        for qi, pos_ids in enumerate(positives):
            pos_rows = [row_of[d] for d in pos_ids if d in row_of]
            if pos_rows:
                pos_vecs = self._index.reconstruct(pos_rows)
                pos_scores = self.student.compute_similarity(q_embs[qi : qi + 1], pos_vecs)[0]
            else:
                pos_scores = np.zeros(0, np.float32)
            keep = [(self._corpus_ids[r], s) for r, s in zip(rows[qi], scores[qi]) if r >= 0 and self._corpus_ids[r] not in set(pos_ids)]
            out.append(select_adversarial([d for d, _ in keep], np.array([s for _, s in keep], np.float32),
                                          pos_scores, self.margin, top_k))
        return out
