"""CPU oracle of the embedding-and-search hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker.  The product (``semantic-search-kd_amd/``)
never imports it; a product path that did would void every parity claim.

Pinning status: the reference keeps no golden vectors for this path and
its engines (faiss-cpu, sentence-transformers) are not installed here, so
  * search  : restates the reference's exact-search idiom (src/kd/eval.py:86,
              tests/conftest.py:184-185) — bit-level outputs are "parity unpinned";
  * encoder : restates HF ``BertModel`` + mean-pool + L2-norm and IS pinned against
              ``transformers.BertModel`` built from an in-memory config in the build
              container (tests/golden/make_golden.py, fixtures in tests/golden/).
"""
