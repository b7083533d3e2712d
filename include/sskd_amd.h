/*
 * sskd_amd.h — C-ABI of the MI355X (gfx950) embedding-and-search path.
 *
 * This is the drop-in boundary for the hot path of Axionis47/semantic-search-kd:
 * everything its `StudentModel` / `FAISSIndexBuilder` wrappers delegate to
 * sentence-transformers and faiss-cpu is reachable through the entry points
 * below.  The reference has no FFI of its own (it is 100 % Python); each entry
 * point therefore cites the reference call site / third-party call it replaces
 * (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add.
 *
 * Conventions (binding):
 *   - plain `extern "C"`, no C++/torch types; pointers + sizes only
 *   - every pointer named `d_*` is a DEVICE pointer (HBM) owned by the caller
 *   - `stream` is a `hipStream_t` passed as `void*` (NULL = default stream);
 *     all work is enqueued on it, nothing synchronises, nothing allocates
 *   - every function returns 0 (SSKD_OK) or an SSKD_ERR_* code and never throws;
 *     `sskd_last_error()` gives a thread-local message for the last failure
 *   - scratch memory is caller-provided; `*_workspace_bytes()` sizes it
 */
#ifndef SSKD_AMD_H
#define SSKD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSKD_ABI_VERSION 1

#define SSKD_OK 0
#define SSKD_ERR_INVALID 1     /* bad argument (null pointer, bad shape, k < 1 ...) */
#define SSKD_ERR_WORKSPACE 2   /* workspace too small */
#define SSKD_ERR_HIP 3         /* a HIP runtime call / kernel launch failed */
#define SSKD_ERR_UNSUPPORTED 4 /* shape outside what the gfx950 kernels are built for */

/* Embedding width the kernels are specialised for (e5-small-v2: configs/kd.yaml:16,
 * src/config.py:30, scripts/build_faiss_index.py:50). */
#define SSKD_DIM 384
/* Corpus rows per HBM tile (one 32x32x2 f32 MFMA M-block). */
#define SSKD_TILE_ROWS 32
/* Largest k served by ONE scan pass; larger k is served by chained passes. */
#define SSKD_K_PASS 32
/* Largest k accepted by sskd_index_search (schemas.py:12-16 caps k at 100 and
 * rerank_top_k at 200). */
#define SSKD_K_MAX 1024

int sskd_abi_version(void);
const char* sskd_last_error(void);
/* Number of HIP devices visible, or -1; does not initialise a context. */
int sskd_device_count(void);

/* ------------------------------------------------------------------------- *
 * Index storage ("add"):  replaces faiss.normalize_L2 + index.add()
 *   reference: scripts/build_faiss_index.py:55-62 (build_from_parquet),
 *              tests/conftest.py:184-185 (IndexFlatIP(384); index.add(x)),
 *              configs/index.yaml:30 (normalize: true)
 * The index lives in HBM as the plain ROW-MAJOR fp32 matrix (1 536 B per row), zero-padded to a multiple of 32
 * rows ("tile" = 32 consecutive rows: lane l of a scanning wave owns row 32t + (l & 31) and the column half
 * 4 (l >> 5) of every 8-column k-step).  The `d_tiled` / `tiled` names of this API date from rounds 1-3, when the
 * tiles were stored in MFMA-fragment order and the screening sidecar carried a second, row-major fp32 copy; callers
 * treat the buffer as opaque (sskd_index_tiled_bytes, sskd_index_add_rows, sskd_index_get_rows).
 * ------------------------------------------------------------------------- */
int64_t sskd_index_padded_rows(int64_t n_rows);
size_t sskd_index_tiled_bytes(int64_t n_rows);

/* Copy `n_rows` row-major fp32 rows into the index starting at index row
 * `dst_row0` (must be a multiple of 32), optionally L2-normalising each row
 * (x / ||x||_2, rows of zero norm left untouched: faiss.normalize_L2).
 * The last partial tile written is zero-padded. */
int sskd_index_add_rows(const float* d_rows, int64_t n_rows, int normalize,
                        float* d_tiled, int64_t dst_row0, void* stream);

/* Inverse of sskd_index_add_rows (for save(): faiss.write_index equivalent). */
int sskd_index_get_rows(const float* d_tiled, int64_t row0, int64_t n_rows,
                        float* d_rows, void* stream);

/* In-place row L2 normalisation (faiss.normalize_L2 on a query batch). */
int sskd_l2_normalize_rows(float* d_x, int64_t n_rows, int dim, void* stream);

/* ------------------------------------------------------------------------- *
 * Exact inner-product top-k ("search"): replaces faiss index.search()
 *   reference: src/serve/app.py:293-301 (index_builder.search(query_emb, k)),
 *              src/kd/eval.py:86 (np.argsort(scores)[::-1][:k]),
 *              tests/conftest.py:184 (IndexFlatIP)
 * d_queries   : row-major fp32 [nq, 384]
 * d_out_scores: fp32 [nq, k], descending; ties broken by lower id
 * d_out_ids   : int64 [nq, k] = local row + id_offset; when fewer than k rows
 *               exist the tail is (-FLT_MAX, -1) like faiss' heap neutral.
 * Scores are the exact fp32 fma chain of the 32x32x2 f32 MFMA in the column
 * order documented in DESIGN.md (oracle/csrc/oracle.c reproduces it bit for bit).
 * ------------------------------------------------------------------------- */
size_t sskd_index_search_workspace_bytes(int64_t n_rows, int nq, int k);
int sskd_index_search(const float* d_tiled, int64_t n_rows,
                      const float* d_queries, int nq, int k, int64_t id_offset,
                      float* d_out_scores, int64_t* d_out_ids,
                      void* d_workspace, size_t workspace_bytes, void* stream);

/* Explicit launch tuning (all zero / NULL = the built-in plan).  The SAME struct must be passed
 * to sskd_index_search_workspace_bytes_ex, sskd_index_search_plan_ex and sskd_index_search_ex:
 * the workspace size depends on it, and there is no process-global (environment) state behind
 * these calls. */
typedef struct sskd_search_tuning {
  int32_t queries_per_block;  /* 0 = auto, else 32 or 64 (B_q of the scan kernel) */
  int32_t target_workgroups;  /* 0 = auto, else the number of workgroups the slicing aims for */
  int32_t pruning_pools;      /* 0 = auto, > 0 force the shared candidate pools on, < 0 off */
} sskd_search_tuning;

size_t sskd_index_search_workspace_bytes_ex(int64_t n_rows, int nq, int k,
                                            const sskd_search_tuning* tuning);
/* sskd_index_search with the tuning struct and two optional hipEvent_t handles (as void*, may be
 * NULL) recorded on `stream` immediately before / after the first scan kernel. */
int sskd_index_search_ex(const float* d_tiled, int64_t n_rows, const float* d_queries, int nq,
                         int k, int64_t id_offset, float* d_out_scores, int64_t* d_out_ids,
                         void* d_workspace, size_t workspace_bytes, void* stream,
                         const sskd_search_tuning* tuning, void* ev_scan_begin, void* ev_scan_end);

/* sskd_index_search with two optional hipEvent_t handles (as void*, may be NULL)
 * recorded on `stream` immediately before and after the first scan kernel: lets a
 * caller time the dominant kernel in-process (bench.py's roofline). */
int sskd_index_search_profiled(const float* d_tiled, int64_t n_rows,
                               const float* d_queries, int nq, int k, int64_t id_offset,
                               float* d_out_scores, int64_t* d_out_ids,
                               void* d_workspace, size_t workspace_bytes, void* stream,
                               void* ev_scan_begin, void* ev_scan_end);

/* Screened search for batch shapes (k <= 10, nq >= 64, >= 2048 rows): a bf16-MFMA screening pass
 * (16x the fp32 matrix rate) over a bf16 copy of the index, then exact fp32 re-scoring of the
 * candidates inside a PROVED error band, so scores and ids are bit-identical to sskd_index_search's.
 * The band is measured, not assumed: |screen - exact| <= |q~ - q| max|row~| + |q| max|row~ - row| +
 * 1e-4 |q| max(max|row|, max|row~|) (Cauchy-Schwarz on the two bf16 roundings + fp32 accumulation slack), with the
 * query norms taken per query and the row maxima when the sidecar is made; in the worst case (every
 * element on a bf16 tie) that is 2^-7 (1 + 2^-9) |q| max|row|, on random data about 0.42 of it.
 * The bf16 copy holds the rows MINUS their mean row (q.mean is the same for every row of a query, so
 * ranking is untouched): on anisotropic embeddings (e5: mean pairwise cosine 0.7-0.8) the centred norms,
 * and with them the band, are 2-2.2x smaller.
 * Every row whose screen score reaches the running bound (a lower bound of the query's k-th best screen
 * score, minus the band) is appended to a per-lane run in the workspace, so the appended set always holds
 * the whole band.  The bound is the minimum of a full pool of K scores of DISTINCT rows: a workgroup whose
 * share of the sample phase lies inside its own slice (slice 0) empties its pool between the two phases, because
 * the same rows are offered again there (round 3 did not, and a row counted twice could lift the bound above
 * the true k-th best score: fixed in round 4, tests/test_screened_gpu.py "count_once" / "topic_sorted").  A query is answered by the exact scan inside the same call only when its band holds
 * more than 256 rows (hundreds of near-duplicates of its neighbours) or one run overflowed (> 64 band rows
 * in one lane's share of a slice) - the in-call fallback is sized for every query, so no output row is
 * ever unproven and callers have nothing to check.  d_status (device int[2]): [0] = always 0 (kept for ABI
 * stability), [1] = the number of queries that took the exact fallback (a cost diagnostic).  ev_scan_begin /
 * ev_scan_end bracket the screening launch (a bound-only sample phase over the shard's first rows, then the slices).  d_bf16: the screening sidecar,
 * sskd_index_bf16_bytes(n_rows) bytes filled by sskd_index_make_bf16 from the CURRENT tiled index
 * (re-make it after sskd_index_add_rows): the bf16 tiles of the centred rows (768 B per row) and a 4-KiB
 * block with max |row|^2, max |row~|^2, max |row~ - (row - mean)|^2 and the column sums.  Exact re-scoring
 * reads the fp32 index itself (row-major since round 4: index + sidecar = 1.5x the corpus; rounds 2-3 kept a
 * second fp32 copy in the sidecar, 2.5x). */
size_t sskd_index_bf16_bytes(int64_t n_rows);
int sskd_index_make_bf16(const float* d_tiled, int64_t n_rows, void* d_bf16, void* stream);
size_t sskd_index_search_screened_workspace_bytes(int64_t n_rows, int nq, int k);
/* launch geometry of the screening pass (roofline accounting): queries per workgroup, corpus passes, slices */
int sskd_index_search_screened_plan(int64_t n_rows, int nq, int k, int* queries_per_block, int* corpus_passes,
                                    int* n_slices);
int sskd_index_search_screened(const float* d_tiled, const void* d_bf16, int64_t n_rows, const float* d_queries,
                               int nq, int k, int64_t id_offset, float* d_out_scores, int64_t* d_out_ids,
                               int* d_status, void* d_workspace, size_t workspace_bytes, void* stream,
                               void* ev_scan_begin, void* ev_scan_end);

/* One-pass variant for the online shape (reference: src/serve/app.py:285-301, schemas.py:12-16 -
 * one query, k <= 100, rerank_top_k <= 200).  sskd_index_search serves k > SSKD_K_PASS by chained
 * corpus passes; this entry point scans the corpus ONCE with plain per-lane lists, collects the best
 * k candidates and PROVES them exact where it can: a row is missing from the candidates only if its
 * list was full and it ranks after that list's last entry, so if the k-th best candidate ranks at or
 * before the best such last entry, the result is the exact top k.  *d_inexact (device int) is set
 * to 0 when every query's result is proven, to 1 otherwise - the caller then falls back to
 * sskd_index_search (results are never silently approximate).  Limits: 1 <= nq <= 64,
 * 1 <= k <= 256, n_rows >= 1.  Outputs as sskd_index_search; stream-ordered, no host sync. */
size_t sskd_index_search_onepass_workspace_bytes(int64_t n_rows, int nq, int k);
int sskd_index_search_onepass(const float* d_tiled, int64_t n_rows, const float* d_queries, int nq,
                              int k, int64_t id_offset, float* d_out_scores, int64_t* d_out_ids,
                              int* d_inexact, void* d_workspace, size_t workspace_bytes,
                              void* stream);

/* Knowledge-distillation losses of the reference and their gradient (loss
 * half).  Replaces MarginMSELoss / ListwiseKDLoss / ContrastiveLoss / CombinedKDLoss.forward
 * (src/kd/losses.py:35-60, 81-106, 127-149, 219-252) on device-resident [batch, n_docs] fp32 score
 * matrices, n_docs <= 64 (document 0 is the positive).  d_losses[4] = { weighted total, margin-MSE,
 * listwise KD, contrastive }; d_grad (may be NULL) = d total / d student, [batch, n_docs];
 * d_row_workspace = 3 * batch floats.  Single components: set the other weights to 0. */
int sskd_kd_loss(const float* d_student, const float* d_teacher, int batch, int n_docs,
                 float temperature, float contrastive_temperature, float w_margin_mse,
                 float w_listwise, float w_contrastive, float* d_losses, float* d_grad,
                 float* d_row_workspace, void* stream);

/* Launch geometry the search would use (for roofline accounting in bench.py):
 * queries per workgroup tile (B_q), corpus passes, slices, waves per workgroup,
 * and scan passes needed for this k. Any out pointer may be NULL. */
int sskd_index_search_plan(int64_t n_rows, int nq, int k, int* queries_per_block,
                           int* corpus_passes, int* n_slices, int* waves_per_block,
                           int* scan_passes);

int sskd_index_search_plan_ex(int64_t n_rows, int nq, int k, const sskd_search_tuning* tuning,
                              int* queries_per_block, int* corpus_passes, int* n_slices,
                              int* waves_per_block, int* scan_passes);

/* Merge `n_lists` per-shard top-k lists per query into one (the step after the
 * RCCL all-gather; the reference has a single index, so no counterpart).
 * d_scores fp32 [n_lists, nq, k_in], d_ids int64 [n_lists, nq, k_in] (global ids,
 * -1 = empty). Output as sskd_index_search. */
int sskd_topk_merge(const float* d_scores, const int64_t* d_ids, int n_lists, int nq,
                    int k_in, int k_out, float* d_out_scores, int64_t* d_out_ids,
                    void* stream);

/* The same merge over PACKED per-shard records, the form one RCCL all-gather moves: record r
 * (r = 0 .. n_lists-1) starts at d_records + r * sskd_topk_record_bytes(nq, k_in) and holds
 * { int64 ids[nq][k_in]; float scores[nq][k_in]; padding to 16 B }.  A rank lets its local search
 * write ids / scores straight into its own record, all-gathers the records once, and merges. */
size_t sskd_topk_record_bytes(int nq, int k);
int sskd_topk_merge_packed(const void* d_records, int n_lists, int nq, int k_in, int k_out,
                           float* d_out_scores, int64_t* d_out_ids, void* stream);

/* q @ d^T in fp32: replaces StudentModel.compute_similarity
 *   reference: src/kd/eval.py:75, tests/test_student_model.py:104-124 */
int sskd_similarity(const float* d_q, int nq, const float* d_d, int nd, int dim,
                    float* d_out, void* stream);

/* ------------------------------------------------------------------------- *
 * Encoder (e5-small-v2 shaped BERT): replaces SentenceTransformer.encode()'s
 * Transformer -> Pooling(mean) -> Normalize modules
 *   reference: tests/test_model_validation.py:80-89,256-262; configs/kd.yaml:16-19
 * ------------------------------------------------------------------------- */

/* Masked mean-pool + L2 normalise:
 *   e = sum_t m_t h_t / max(sum_t m_t, 1e-9);  if normalize: e /= max(||e||, 1e-12)
 * d_hidden: [B, S, 384] bf16 (hidden_is_bf16 = 1) or fp32 (0); d_mask int32 [B, S];
 * d_out fp32 [B, 384] row-major. */
int sskd_pool_normalize(const void* d_hidden, int hidden_is_bf16, const int32_t* d_mask,
                        int B, int S, int normalize, float* d_out, void* stream);

/* Architecture of the bi-encoder (src/config.py:22-32). */
typedef struct sskd_encoder_config {
  int32_t vocab_size;        /* 30522 */
  int32_t hidden;            /* 384 (only value supported) */
  int32_t layers;            /* 12 */
  int32_t heads;             /* 12 (head dim 32, only value supported) */
  int32_t intermediate;      /* 1536 (only value supported) */
  int32_t max_positions;     /* 512 */
  int32_t type_vocab;        /* 2 */
  float layer_norm_eps;      /* 1e-12 */
} sskd_encoder_config;

/* Device weight table.  All matrices are bf16 row-major in the HF nn.Linear
 * convention [out_features, in_features]; biases and LayerNorm parameters fp32. */
typedef struct sskd_encoder_layer_weights {
  const void* wqkv;   /* bf16 [1152, 384]  (Wq; Wk; Wv stacked) */
  const float* bqkv;  /* [1152] */
  const void* wo;     /* bf16 [384, 384] */
  const float* bo;    /* [384] */
  const float* ln1_g; /* [384] */
  const float* ln1_b;
  const void* w1;     /* bf16 [1536, 384] */
  const float* b1;    /* [1536] */
  const void* w2;     /* bf16 [384, 1536], tiled chunk-major for the fused MLP (see weights.py) */
  const float* b2;    /* [384] */
  const float* ln2_g;
  const float* ln2_b;
} sskd_encoder_layer_weights;

typedef struct sskd_encoder_weights {
  const void* word_emb;  /* bf16 [vocab, 384] */
  const void* pos_emb;   /* bf16 [max_positions, 384] */
  const void* type_emb;  /* bf16 [type_vocab, 384] (row 0 is used) */
  const float* emb_ln_g; /* [384] */
  const float* emb_ln_b;
  const sskd_encoder_layer_weights* layers; /* HOST array of `layers` entries */
} sskd_encoder_weights;

size_t sskd_encoder_workspace_bytes(const sskd_encoder_config* cfg, int B, int S);

/* Full forward: ids/mask int32 [B, S] -> L2-normalised (if normalize) fp32 [B, 384].
 * bf16 activations and MFMA operands, fp32 accumulation / LayerNorm / softmax.
 * Stream semantics: everything is ordered after the work already in `stream` and complete for work enqueued on it
 * afterwards.  Batches whose halves still fill the chip twice (>= 2 x 65 536 padded tokens) run as two halves, one of
 * them on an internal side stream forked from and joined back into `stream` with events (legal under stream capture: the
 * graph gets two branches); the result is bit-identical, rows do not interact.  SSKD_FORWARD_STREAMS=1 in the environment
 * keeps every forward of the library (this one, the packed one, sskd_teacher_score) on the caller's stream alone. */
int sskd_encoder_forward(const sskd_encoder_config* cfg, const sskd_encoder_weights* w,
                         const int32_t* d_ids, const int32_t* d_mask, int B, int S,
                         int normalize, float* d_out, void* d_workspace,
                         size_t workspace_bytes, void* stream);

/* Same forward, stopping after the last encoder layer: bf16 [B, S, 384] hidden
 * states (test hook for per-stage parity against the oracle). */
int sskd_encoder_hidden(const sskd_encoder_config* cfg, const sskd_encoder_weights* w,
                        const int32_t* d_ids, const int32_t* d_mask, int B, int S,
                        void* d_hidden_bf16, void* d_workspace, size_t workspace_bytes,
                        void* stream);

/* ------------------------------------------------------------------------- *
 * Sequence packing (varlen / "cu_seqlens"): SentenceTransformer.encode pads every text of a
 * batch to the longest one (reference: tests/test_model_validation.py:80-110); here whole
 * sequences are concatenated into rows of `capacity` tokens (<= 256, multiple of 32) and
 * attention is block-diagonal per sequence, so only the tail of a row is padding.
 * ------------------------------------------------------------------------- */

/* HOST function.  Best-fit-decreasing placement of n_seq sequences (1 <= lengths[i] <= capacity)
 * into rows.  table[4 i .. 4 i + 3] = { row, lo, hi, i }: sequence i occupies tokens [lo, hi) of
 * `row`.  *n_rows = rows used. */
int sskd_pack_plan(const int32_t* lengths, int n_seq, int capacity, int32_t* table, int* n_rows);

/* Lay the flat token stream out as packed rows: d_flat_ids holds the sequences back to back,
 * d_cu_seqlens[n_seq + 1] their start offsets, d_table the (device copy of the) plan.
 * Outputs int32 [n_rows, capacity]: d_ids (0 = [PAD]) and d_seg (lo | hi << 16 per token,
 * 0 for padding). */
int sskd_pack_tokens(const int32_t* d_flat_ids, const int32_t* d_cu_seqlens, const int32_t* d_table,
                     int n_seq, int n_rows, int capacity, int32_t* d_ids, int32_t* d_seg, void* stream);

/* sskd_encoder_forward over packed rows; position ids restart at every sequence.  d_out fp32
 * [*, 384]: row table[4 i + 3] receives the embedding of sequence i.  Workspace:
 * sskd_encoder_workspace_bytes(cfg, n_rows, capacity). */
int sskd_encoder_forward_packed(const sskd_encoder_config* cfg, const sskd_encoder_weights* w,
                                const int32_t* d_ids, const int32_t* d_seg, int n_rows, int capacity,
                                const int32_t* d_table, int n_seq, int normalize, float* d_out,
                                void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Dimension-generic post-LN BERT-family encoder with SAVED activations and its backward pass:
 * the student forward/backward of the KD training step (reference: src/kd/train.py:176-210,
 * StudentModel.encode_with_gradients -> torch autograd) and the forward of the teacher
 * cross-encoder (reference: src/mining/miners.py:128-151, src/serve/app.py:321-339).
 * Row-major activations; hidden <= 1024; head width, hidden, intermediate multiples of 32;
 * S a multiple of 32 (pad with mask 0).  Weights bf16 row-major [out, in]; the `_t` members are
 * the transposes [in, out] (needed by the backward pass only, may be NULL for inference).
 * ------------------------------------------------------------------------- */
typedef struct sskd_generic_config {
  int32_t vocab_size;
  int32_t hidden;
  int32_t layers;
  int32_t heads;
  int32_t intermediate;
  int32_t max_positions;
  int32_t type_vocab;
  float layer_norm_eps;
  int32_t pos_offset;     /* position id of token 0: 0 (BERT), 2 (XLM-R: padding_idx + 1) */
} sskd_generic_config;

typedef struct sskd_generic_layer_weights {
  const void* wqkv;   /* bf16 [3H, H] (Wq; Wk; Wv) */
  const void* wqkv_t; /* bf16 [H, 3H] */
  const void* wo;     /* bf16 [H, H] */
  const void* wo_t;
  const void* w1;     /* bf16 [F, H] */
  const void* w1_t;   /* bf16 [H, F] */
  const void* w2;     /* bf16 [H, F] */
  const void* w2_t;   /* bf16 [F, H] */
  const float* bqkv;  /* [3H] */
  const float* bo;
  const float* ln1_g;
  const float* ln1_b;
  const float* b1;    /* [F] */
  const float* b2;
  const float* ln2_g;
  const float* ln2_b;
} sskd_generic_layer_weights;

typedef struct sskd_generic_weights {
  const void* word_emb;  /* bf16 [vocab, H] */
  const void* pos_emb;   /* bf16 [max_positions, H] */
  const void* type_emb;  /* bf16 [type_vocab, H] (row 0 is used) */
  const float* emb_ln_g;
  const float* emb_ln_b;
  const sskd_generic_layer_weights* layers; /* HOST array */
} sskd_generic_weights;

/* fp32 gradient buffers with the shapes of the parameters; the backward pass ADDS into them. */
typedef struct sskd_generic_layer_grads {
  float* wqkv;  /* [3H, H] */
  float* bqkv;
  float* wo;
  float* bo;
  float* ln1_g;
  float* ln1_b;
  float* w1;
  float* b1;
  float* w2;
  float* b2;
  float* ln2_g;
  float* ln2_b;
} sskd_generic_layer_grads;

typedef struct sskd_generic_grads {
  float* word_emb;
  float* pos_emb;
  float* type_emb;  /* row 0 receives the gradient */
  float* emb_ln_g;
  float* emb_ln_b;
  const sskd_generic_layer_grads* layers; /* HOST array */
} sskd_generic_grads;

/* training != 0: room for every layer's saved activations + backward scratch. */
size_t sskd_generic_workspace_bytes(const sskd_generic_config* cfg, int B, int S, int training);

/* Forward.  pool != 0: masked mean-pool (+ L2 normalise) -> d_out fp32 [B, H]; pool == 0: d_out
 * receives the final hidden states bf16 [B, S, H].  With training != 0 the workspace afterwards
 * holds what sskd_generic_backward needs (same B, S, same workspace). */
int sskd_generic_forward(const sskd_generic_config* cfg, const sskd_generic_weights* w, const int32_t* d_ids,
                         const int32_t* d_mask, int B, int S, int training, int pool, int normalize, void* d_out,
                         void* d_workspace, size_t workspace_bytes, void* stream);

/* Backward of the pooled forward: d_dout fp32 [B, H] = d loss / d embeddings. */
int sskd_generic_backward(const sskd_generic_config* cfg, const sskd_generic_weights* w, const sskd_generic_grads* grads,
                          const int32_t* d_ids, const int32_t* d_mask, int B, int S, int normalize, const float* d_dout,
                          void* d_workspace, size_t workspace_bytes, void* stream);

/* Teacher cross-encoder score (replaces CrossEncoder.predict behind TeacherModel.score; reference:
 * src/mining/miners.py:135-137, src/serve/app.py:325-326; model = XLM-R-large shaped
 * XLMRobertaForSequenceClassification with one label, docs/adr-002): generic encoder forward ->
 * hidden state of token 0 -> dense [H, H] + tanh -> out_proj [1, H] -> d_logits fp32 [B] (raw logits).
 * The head runs in fp32: head weights fp32 row-major [out, in], biases fp32 (a reranker's product is the
 * ORDER of its logits; the head is 0.002 % of the FLOPs). */
size_t sskd_teacher_workspace_bytes(const sskd_generic_config* cfg, int B, int S);
int sskd_teacher_score(const sskd_generic_config* cfg, const sskd_generic_weights* w, const float* d_head_dense_w,
                       const float* d_head_dense_b, const float* d_head_out_w, const float* d_head_out_b,
                       const int32_t* d_ids, const int32_t* d_mask, int B, int S, float* d_logits, void* d_workspace,
                       size_t workspace_bytes, void* stream);

/* C[M, N] (bf16 or fp32, optionally +=) = A[M, K] . B[N, K]^T + bias[N]: the NT GEMM every
 * product of the generic path goes through (test hook; K % 32 == 0). */
int sskd_gemm_nt_bf16(const void* d_a, const void* d_b, void* d_c, const float* d_bias, int M, int N, int K,
                      int c_is_f32, int accumulate, void* stream);
/* Which kernels serve the PLAIN large products of the generic path (C = A . W^T + bias, bf16 out, K and N >= 1024: the
 * teacher's QKV, attention-output and FFN2 products): 0 = automatic - hipBLASLt, whose tuned kernels are 15-25 % faster
 * there than this library's 256 x 256 MFMA kernel; 1 = this library's kernels only (tests, A/B probes).  Any other value
 * only queries.  Process-wide; returns the mode in force.  Products with a fused epilogue, K = 384 (student) products,
 * batched / fp32 / accumulating products always run on the hand-written kernels. */
int sskd_gemm_backend(int mode);
/* C[M, N] (fp32) += A[T, M]^T . B[T, N]: the weight-gradient product of the training step with the token
 * dimension as the row of both bf16 operands (test hook; M % 384 == 0, N % 128 == 0, T % 64 == 0, else
 * SSKD_ERR_UNSUPPORTED: the step then transposes and uses the NT kernel). */
int sskd_gemm_tn_bf16(const void* d_a, const void* d_b, float* d_c, int64_t T, int M, int N, void* stream);

/* ------------------------------------------------------------------------- *
 * Host WordPiece tokenizer (uncased BERT): replaces the `tokenizers` call inside
 * SentenceTransformer.encode for ASCII text (reference: AutoTokenizer use src/utils/chunk.py:26;
 * vocabulary / special ids SURVEY.md §8c).  HOST functions, multi-threaded, no device work.
 * ------------------------------------------------------------------------- */

/* vocab_blob: the vocabulary tokens in id order separated by '\n' (vocab.txt layout). */
int sskd_tokenizer_create(const char* vocab_blob, int64_t blob_bytes, void** handle);
void sskd_tokenizer_destroy(void* handle);

/* Tokenise n_texts UTF-8 texts: text i = text_blob[offsets[i], offsets[i+1]) (NUL bytes inside are
 * ignored, so texts may simply be NUL-joined).  Emits "[CLS] pieces [SEP]" ids truncated to
 * max_len (the closing [SEP] is kept) back to back into out_ids; out_lengths[i] = ids of text i.
 * A text containing any non-ASCII byte is not tokenised: out_needs_unicode[i] = 1,
 * out_lengths[i] = 0 - the caller must route it through a Unicode-complete tokenizer.
 * *out_total = ids written; SSKD_ERR_WORKSPACE if out_capacity is too small. */
int sskd_tokenizer_encode(void* handle, const char* text_blob, const int64_t* offsets, int n_texts,
                          int max_len, int n_threads, int32_t* out_ids, int64_t out_capacity,
                          int32_t* out_lengths, uint8_t* out_needs_unicode, int64_t* out_total);

#ifdef __cplusplus
}
#endif
#endif /* SSKD_AMD_H */
