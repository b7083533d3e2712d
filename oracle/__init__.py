"""CPU oracle of the embedding-and-search hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker.  The product (``semantic-search-kd_amd/``)
never imports it; a product path that did would void every parity claim.

Pinning status: the reference keeps no golden vectors for this path and its
engines (faiss-cpu, sentence-transformers) are not installed here; every oracle is pinned by output
of reference-held or reference-executed code generated in the build container
(``tests/golden/make_golden.py``, fixtures under ``tests/golden/``):
  * search  : ``search_ref_*.npz`` — ids / scores computed by the reference's own
              ``scripts/simple_eval.py::evaluate_model`` and ``src/kd/eval.py::KDEvaluator``;
  * encoder : ``bert_*.npz`` — ``transformers.BertModel`` from an in-memory config (forward, benign and
              stress weights) and its torch-autograd gradients (``bert_grads_small.npz``);
  * KD loss : ``kd_loss.npz`` — the reference's own ``src/kd/losses.py``;
  * teacher : ``xlmr_small.npz`` — ``transformers.XLMRobertaForSequenceClassification``;
  * mining  : ``ance_mining.json`` — the reference's own ``src/mining/miners.py::ANCEMiner``.
"""
