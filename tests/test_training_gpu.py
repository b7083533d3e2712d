"""Student forward / backward (BASELINE cfg 4) on the GPU vs torch autograd over the fp32 oracle.

Gradient oracle: ``oracle.encoder.embeddings_torch`` under torch autograd on the CPU; that restatement's
gradients are pinned to ``transformers.BertModel`` autograd by tests/golden/bert_grads_small.npz
(make_golden.make_bert_grads; the CPU test below).  Tolerance (bf16 compute, fp32 accumulation):
cosine >= 0.999 per parameter tensor, norms within 3 %.
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import encoder as enc_oracle
from oracle import kd_losses as kd_oracle
from semantic_search_kd_amd import BertConfig, synthetic_state_dict


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def _oracle_grads(sd, cfg, ids, mask, probe):
    t = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd.items()}
    e = enc_oracle.embeddings_torch(t, ids, mask, cfg.num_hidden_layers, cfg.num_attention_heads)
    (e * torch.from_numpy(probe)).sum().backward()
    return e.detach().numpy(), {k: v.grad.numpy() for k, v in t.items()}


def test_oracle_autograd_is_pinned_by_transformers_fixture():
    """CPU: torch autograd over oracle/encoder.py reproduces the committed transformers.BertModel gradients."""
    import sys

    sys.path.insert(0, str(GOLDEN))
    from make_golden import grad_case

    gold = np.load(GOLDEN / "bert_grads_small.npz")
    cfg, sd, ids, mask, probe = grad_case()
    emb, grads = _oracle_grads(sd, cfg, ids, mask, probe)
    np.testing.assert_allclose(emb, gold["embeddings"], atol=1e-5)
    names = [k[5:] for k in gold.files if k.startswith("grad:")]
    assert len(names) == 37
    top = max(float(gold["amax:" + k]) for k in names)
    for k in names:
        want = gold["grad:" + k].astype(np.float64) * float(gold["amax:" + k])
        if float(gold["amax:" + k]) < 1e-6 * top:
            # mathematically zero (a key bias shifts every score of a query alike: softmax ignores it)
            assert np.abs(grads[k]).max() < 1e-6 * top and "key.bias" in k
            continue
        assert _cos(grads[k], want) > 0.99999, k
        assert abs(np.linalg.norm(grads[k]) / np.linalg.norm(want) - 1.0) < 2e-3, k


@pytest.mark.gpu
def test_nt_gemm_matches_torch(gpu, native_lib):
    from semantic_search_kd_amd import _native

    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K, f32, acc in ((128, 128, 64, 0, 0), (300, 130, 96, 1, 0), (1000, 384, 384, 0, 0), (77, 1536, 32, 1, 1),
                               (384, 1152, 4096, 1, 1), (5, 3, 32, 0, 0),
                               # the 256 x 256 global-to-LDS kernel (M, N multiples of 256, K of 64, bf16 out)
                               (1024, 256, 64, 0, 0), (2048, 768, 1024, 0, 0), (1280, 512, 192, 0, 0), (1536, 384, 384, 0, 0),
                               (4096, 1152, 128, 0, 0), (2304, 1280, 64, 0, 0)):
        a = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
        b = torch.randn((N, K), generator=g, device="cuda").to(torch.bfloat16)
        bias = torch.randn(N, generator=g, device="cuda")
        c0 = torch.randn((M, N), generator=g, device="cuda") if acc else None
        c = (c0.clone() if acc else torch.empty((M, N), device="cuda", dtype=torch.float32 if f32 else torch.bfloat16))
        _native.check(native_lib.sskd_gemm_nt_bf16(a.data_ptr(), b.data_ptr(), c.data_ptr(), bias.data_ptr(), M, N, K, f32, acc,
                                                   int(torch.cuda.current_stream().cuda_stream)))
        want = a.float() @ b.float().T + bias + (c0 if acc else 0)
        tol = 2e-2 * want.abs().max().item() if not f32 else 1e-3 * max(want.abs().max().item(), 1.0)
        assert (c.float() - want).abs().max().item() <= tol, (M, N, K)


@pytest.mark.gpu
def test_plain_large_k_products_library_route_matches_own_kernel(gpu, native_lib):
    """csrc/blaslt.hip: the teacher's plain products (K, N >= 1024, bias only, bf16 out) go through hipBLASLt in
    automatic mode and through the 256 x 256 MFMA kernel with sskd_gemm_backend(1); both against torch in fp32, with
    and without a bias, with a leading dimension larger than the row (the QKV output is consumed as three column
    blocks), and the switch itself (query, set, restore)."""
    from semantic_search_kd_amd import _native

    lib = native_lib
    assert lib.sskd_gemm_backend(-1) == 0           # default: automatic
    g = torch.Generator(device="cuda").manual_seed(1)
    st = int(torch.cuda.current_stream().cuda_stream)
    try:
        for M, N, K, use_bias in ((2048, 3072, 1024, True), (1024, 1024, 1024, True), (2048, 1024, 4096, True),
                                  (4096, 1024, 1024, False), (1024, 2048, 2048, True)):
            a = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
            b = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, generator=g, device="cuda")
            want = a.float() @ b.float().T + (bias if use_bias else 0)
            got = {}
            for mode in (1, 0):
                assert lib.sskd_gemm_backend(mode) == mode
                c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
                _native.check(lib.sskd_gemm_nt_bf16(a.data_ptr(), b.data_ptr(), c.data_ptr(), bias.data_ptr() if use_bias else None,
                                                    M, N, K, 0, 0, st))
                torch.cuda.synchronize()
                err = (c.float() - want).abs().max().item()
                assert err <= 2e-2 * want.abs().max().item(), (mode, M, N, K, err)
                got[mode] = c
            # two correct bf16 roundings of the same product: equal almost everywhere, never more than an ulp or two apart
            d = (got[0].float() - got[1].float()).abs()
            assert d.max().item() <= 4e-2 * want.abs().max().item() and (d > 0).float().mean().item() < 0.2
    finally:
        lib.sskd_gemm_backend(0)


@pytest.mark.gpu
def test_tn_gemm_matches_torch(gpu, native_lib):
    """C += A^T B with the token dimension as the row of both operands (transposing LDS reads, split over K)."""
    from semantic_search_kd_amd import _native

    g = torch.Generator(device="cuda").manual_seed(1)
    st = int(torch.cuda.current_stream().cuda_stream)
    for T, M, N in ((64, 384, 128), (4096, 384, 384), (8192, 1152, 384), (2048, 1536, 384), (1024, 384, 1536), (320, 768, 256)):
        a = torch.randn((T, M), generator=g, device="cuda").to(torch.bfloat16)
        b = torch.randn((T, N), generator=g, device="cuda").to(torch.bfloat16)
        c0 = torch.randn((M, N), generator=g, device="cuda")
        c = c0.clone()
        _native.check(native_lib.sskd_gemm_tn_bf16(a.data_ptr(), b.data_ptr(), c.data_ptr(), T, M, N, st))
        want = a.float().T @ b.float() + c0
        assert (c - want).abs().max().item() <= 1e-3 * max(want.abs().max().item(), 1.0), (T, M, N)
    # unsupported shapes are refused, not mis-computed
    a = torch.zeros((64, 256), device="cuda", dtype=torch.bfloat16)
    c = torch.zeros((256, 128), device="cuda")
    assert native_lib.sskd_gemm_tn_bf16(a.data_ptr(), a.data_ptr(), c.data_ptr(), 64, 256, 128, st) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("dims", ["small", "e5", "heads64", "long", "unfused"])
def test_encoder_gradients_match_oracle_autograd(gpu, dims):
    from semantic_search_kd_amd.training import TrainableEncoder

    if dims == "small":
        cfg = BertConfig(vocab_size=600, hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                         intermediate_size=512, max_position_embeddings=64)
        ids, mask = enc_oracle.synthetic_token_ids(5, 40, seed=31, vocab=600, lengths=[40, 33, 17, 8, 2])
    elif dims == "heads64":  # head width 64: the other instantiation of the fused attention forward / backward
        cfg = BertConfig(vocab_size=600, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                         intermediate_size=256, max_position_embeddings=128)
        ids, mask = enc_oracle.synthetic_token_ids(4, 128, seed=35, vocab=600, lengths=[128, 97, 64, 3])
    elif dims == "long":     # 256 tokens: all eight waves of the fused attention kernels own a tile
        cfg = BertConfig(vocab_size=600, hidden_size=64, num_hidden_layers=1, num_attention_heads=2,
                         intermediate_size=128, max_position_embeddings=256)
        ids, mask = enc_oracle.synthetic_token_ids(3, 256, seed=36, vocab=600, lengths=[256, 200, 33])
    elif dims == "unfused":  # S * head width beyond the fused backward's LDS budget: materialised-score path
        cfg = BertConfig(vocab_size=600, hidden_size=128, num_hidden_layers=1, num_attention_heads=2,
                         intermediate_size=256, max_position_embeddings=256)
        ids, mask = enc_oracle.synthetic_token_ids(2, 160, seed=37, vocab=600, lengths=[160, 90])
    else:  # the e5-small-v2 architecture, 2 layers, reduced vocabulary (the embedding table is a gather)
        cfg = BertConfig(vocab_size=2000, num_hidden_layers=2)
        ids, mask = enc_oracle.synthetic_token_ids(6, 70, seed=33, vocab=2000, lengths=[70, 64, 33, 32, 9, 2])
    sd = synthetic_state_dict(cfg)
    probe = np.random.Generator(np.random.PCG64(5)).standard_normal((ids.shape[0], cfg.hidden_size)).astype(np.float32)
    want_e, want_g = _oracle_grads(sd, cfg, ids, mask, probe)
    model = TrainableEncoder(cfg, sd, "cuda:0")
    emb = model(ids, mask, normalize=True)
    assert emb.requires_grad and emb.shape == want_e.shape
    e = emb.detach().cpu().numpy()
    assert min(_cos(e[i], want_e[i]) for i in range(e.shape[0])) >= 0.999
    (emb * torch.from_numpy(probe).cuda()).sum().backward()
    top = max(np.abs(v).max() for v in want_g.values())
    for name in model.names:
        got = model.p(name).grad.cpu().numpy()
        ref = want_g[name]
        if np.abs(ref).max() < 1e-6 * top:  # mathematically zero (key biases): only rounding noise may show
            assert np.abs(got).max() < 2e-2 * top, name
            continue
        assert _cos(got, ref) >= 0.999, (name, _cos(got, ref))
        assert abs(np.linalg.norm(got) / np.linalg.norm(ref) - 1.0) < 0.03, name
    # rows of the embedding tables that no token touched receive exactly zero gradient
    gw = model.p("embeddings.word_embeddings.weight").grad
    untouched = torch.ones(cfg.vocab_size, dtype=torch.bool)
    untouched[torch.from_numpy(ids[mask.astype(bool)]).long()] = False
    assert not gw[untouched.cuda()].any()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["stress_l2", "stress_l12"])
def test_trainable_forward_on_stress_weights_matches_transformers_golden(gpu, tag):
    """The generic row-major path (256-tile GEMMs, fused attention with saved log-sum-exp) on the hard-case
    weights - peaky softmax rows, LayerNorm gains in [0.3, 3], +-10 outlier channels - against the committed
    transformers.BertModel embeddings: the same gate as the specialised inference kernels
    (tests/test_encoder_gpu.py), so the training forward is not only exercised on benign weights."""
    from conftest import GOLDEN
    from semantic_search_kd_amd.training import TrainableEncoder

    gold = np.load(GOLDEN / f"bert_{tag}.npz")
    cfg = BertConfig(num_hidden_layers=int(gold["layers"]))
    model = TrainableEncoder(cfg, synthetic_state_dict(cfg, stress=True), "cuda:0")
    with torch.no_grad():
        emb = model(gold["input_ids"], gold["attention_mask"], normalize=True).cpu().numpy()
    cos = [_cos(emb[i], gold["embeddings"][i]) for i in range(emb.shape[0])]
    assert min(cos) >= 0.999, cos


@pytest.mark.gpu
def test_kd_training_step_matches_oracle_and_learns(gpu):
    """The reference's step (src/kd/train.py:176-210) on the HIP path: encode_with_gradients twice ->
    q @ d.T -> CombinedKDLoss (HIP) -> backward -> AdamW; loss and gradients vs the oracle chain
    (oracle encoder autograd + the reference-pinned KD-loss oracle), then a few steps lower the loss."""
    from semantic_search_kd_amd import CombinedKDLoss, StudentModel
    from semantic_search_kd_amd.encoder import Mi355xSentenceEncoder

    cfg = BertConfig(vocab_size=2000, num_hidden_layers=2)
    sd = synthetic_state_dict(cfg)
    enc = Mi355xSentenceEncoder(None, "cuda:0", config=cfg, state_dict=sd)
    student = StudentModel.from_encoder(enc, "e5-small-v2-synthetic")
    q_ids, q_mask = enc_oracle.synthetic_token_ids(1, 12, seed=41, vocab=2000)
    d_ids, d_mask = enc_oracle.synthetic_token_ids(9, 60, seed=42, vocab=2000, lengths=[60, 55, 41, 33, 32, 20, 11, 7, 3])
    teacher = np.random.Generator(np.random.PCG64(43)).standard_normal(9).astype(np.float32) * 3.0
    model = enc.trainable()
    student.model.train()
    loss_fn = CombinedKDLoss()

    def step_loss():
        q = model(q_ids, q_mask)
        d = model(d_ids, d_mask)
        scores = torch.matmul(q, d.T)[0]
        return loss_fn(scores.unsqueeze(0), torch.from_numpy(teacher).cuda().unsqueeze(0)), scores

    out, scores = step_loss()
    out["loss"].backward()
    # oracle chain
    t = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd.items()}
    qo = enc_oracle.embeddings_torch(t, q_ids, q_mask, 2)
    do = enc_oracle.embeddings_torch(t, d_ids, d_mask, 2)
    so = torch.matmul(qo, do.T)[0]
    ref, ref_ds = kd_oracle.combined(so.detach().numpy()[None], teacher[None], loss_fn.current_temperature)
    so.backward(gradient=torch.from_numpy(ref_ds[0].astype(np.float32)))
    assert np.abs(scores.detach().cpu().numpy() - so.detach().numpy()).max() < 5e-3
    assert abs(float(out["loss"]) - ref["loss"]) < 2e-2 * max(1.0, abs(ref["loss"]))
    checked = 0
    for name in model.names:
        refg = t[name].grad.numpy()
        if "key.bias" in name or np.linalg.norm(refg) < 1e-10:
            continue
        c = _cos(model.p(name).grad.cpu().numpy(), refg)
        assert c >= 0.999, (name, c)   # the per-tensor gate of the encoder-only test holds for the whole step too
        checked += 1
    assert checked >= 30
    # optimisation: the same batch, a few AdamW steps
    opt = torch.optim.AdamW(student.model.parameters(), lr=2e-4)
    first = None
    for _ in range(8):
        opt.zero_grad()
        out, _ = step_loss()
        out["loss"].backward()
        opt.step()
        first = float(out["loss"]) if first is None else first
    assert float(out["loss"]) < first
    # inference weights follow the trained masters on the next encode
    before = enc.encode_token_ids(d_ids, d_mask).cpu().numpy()
    want = model(d_ids, d_mask).detach().cpu().numpy()
    assert min(_cos(before[i], want[i]) for i in range(9)) > 0.999


@pytest.mark.gpu
def test_kd_step_at_bench_depth_and_geometry_vs_oracle(gpu):
    """BASELINE cfg 4 as bench.py runs it - the full e5-small-v2 architecture (12 layers), queries of 32 tokens and
    passages of 256 tokens, (query, positive, 7 negatives) tuples, two encodes -> q . d -> CombinedKDLoss (HIP) ->
    backward INTO the flat gradient buffer - with 4 tuples instead of 32 so that the CPU oracle chain (oracle encoder
    under torch autograd + the reference-pinned KD-loss oracle) finishes in seconds.  Loss, scores and every
    parameter gradient are compared; the accumulation of the two backward calls into one ``.grad`` is part of it."""
    from semantic_search_kd_amd import CombinedKDLoss
    from semantic_search_kd_amd.training import TrainableEncoder

    cfg = BertConfig(vocab_size=3000)
    assert cfg.num_hidden_layers == 12 and cfg.hidden_size == 384
    sd = synthetic_state_dict(cfg)
    tuples, docs = 4, 8
    q_ids, q_mask = enc_oracle.synthetic_token_ids(tuples, 32, seed=51, vocab=3000, lengths=[32, 20, 9, 5])
    d_ids, d_mask = enc_oracle.synthetic_token_ids(tuples * docs, 256, seed=52, vocab=3000,
                                                   lengths=[256, 200, 131, 90, 77, 64, 33, 12] * tuples)
    teacher = np.random.Generator(np.random.PCG64(53)).standard_normal((tuples, docs)).astype(np.float32) * 3.0
    model = TrainableEncoder(cfg, sd, "cuda:0")
    loss_fn = CombinedKDLoss()
    q = model(q_ids, q_mask)
    d = model(d_ids, d_mask).view(tuples, docs, -1)
    scores = torch.einsum("th,tdh->td", q, d)
    out = loss_fn(scores, torch.from_numpy(teacher).cuda())
    out["loss"].backward()
    grads = {n: model.p(n).grad.detach().cpu().numpy() for n in model.names}
    # the gradients live in ONE flat buffer and both backward calls accumulated into it
    assert all(model.p(n).grad.untyped_storage().data_ptr() == model._flat_grad.untyped_storage().data_ptr() for n in model.names)
    t = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd.items()}
    qo = enc_oracle.embeddings_torch(t, q_ids, q_mask, 12)
    do = enc_oracle.embeddings_torch(t, d_ids, d_mask, 12).view(tuples, docs, -1)
    so = torch.einsum("th,tdh->td", qo, do)
    ref, ref_ds = kd_oracle.combined(so.detach().numpy(), teacher, loss_fn.current_temperature)
    so.backward(gradient=torch.from_numpy(ref_ds.astype(np.float32)))
    assert np.abs(scores.detach().cpu().numpy() - so.detach().numpy()).max() < 1e-2
    assert abs(float(out["loss"].detach()) - ref["loss"]) < 2e-2 * max(1.0, abs(ref["loss"]))
    worst, checked = 1.0, 0
    for name in model.names:
        refg = t[name].grad.numpy()
        if "key.bias" in name or np.linalg.norm(refg) < 1e-10:
            continue
        c = _cos(grads[name], refg)
        worst = min(worst, c)
        assert c >= 0.999, (name, c)   # measured worst 0.99954 over 185 tensors (gpurun_out/r03_prof: 12 layers)
        checked += 1
    print(f"12-layer KD step: {checked} parameter tensors, worst gradient cosine {worst:.5f}")
    assert checked >= 150
    # a second step after zero_grad(set_to_none=True) starts from a zeroed buffer again (no stale accumulation)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-5)
    opt.zero_grad(set_to_none=True)
    assert model.p(model.names[0]).grad is None
    out2 = loss_fn(torch.einsum("th,tdh->td", model(q_ids, q_mask), model(d_ids, d_mask).view(tuples, docs, -1)),
                   torch.from_numpy(teacher).cuda())
    out2["loss"].backward()
    for name in ("encoder.layer.11.output.dense.weight", "embeddings.word_embeddings.weight"):
        assert _cos(model.p(name).grad.cpu().numpy(), grads[name]) > 0.9999, name


@pytest.mark.gpu
def test_graphed_kd_step_replays_the_eager_step(gpu):
    """training.GraphedStep: the whole KD step (two encodes, q . d, the fused loss, backward, AdamW) captured into ONE
    HIP graph.  Replayed for a few steps on changing batches it follows the eager step (reference step:
    src/kd/train.py:176-210): same losses to 1e-3 relative and parameters within the tolerance that the
    fp32-atomic weight-gradient sums leave between two eager runs."""
    from semantic_search_kd_amd import BertConfig, synthetic_state_dict
    from semantic_search_kd_amd.bench_support import synthetic_ids
    from semantic_search_kd_amd.losses import CombinedKDLoss
    from semantic_search_kd_amd.training import GraphedStep, TrainableEncoder

    cfg = BertConfig(num_hidden_layers=2)
    dev = torch.device("cuda:0")
    tuples, docs = 4, 8
    batches = []
    for s in range(4):
        q_ids, q_mask = synthetic_ids(tuples, 32, cfg.vocab_size, dev, seed=10 + s)
        d_ids, d_mask = synthetic_ids(tuples * docs, 64, cfg.vocab_size, dev, seed=20 + s)
        teacher = torch.randn((tuples, docs), generator=torch.Generator(device=dev).manual_seed(30 + s), device=dev) * 3.0
        batches.append((q_ids, q_mask, d_ids, d_mask, teacher))

    def run(graphed: bool):
        model = TrainableEncoder(cfg, synthetic_state_dict(cfg), dev)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
        loss_fn = CombinedKDLoss()
        static = [t.clone() for t in batches[0]]

        def step():
            opt.zero_grad(set_to_none=True)
            q = model(static[0], static[1])
            d = model(static[2], static[3]).view(tuples, docs, -1)
            out = loss_fn(torch.einsum("th,tdh->td", q, d), static[4])
            out["loss"].backward()
            opt.step()
            return out["loss"].detach()

        losses = []
        if graphed:
            # the capture's two warm-up steps train on batch 0 (the captured step itself is only recorded, not run):
            # the eager arm takes the same two steps
            runner = GraphedStep(step, warmup=2, modules=[model])
        else:
            for _ in range(2):
                step()
            runner = step
        for b in batches:
            GraphedStep.copy_inputs(static, b)
            losses.append(float(runner()))
        # an EAGER forward after the replays must see the trained weights (a replay does not bump torch's version
        # counters: GraphedStep(modules=...) marks the bf16 device copies stale instead)
        emb = model(batches[0][0], batches[0][1]).detach().cpu().numpy()
        torch.cuda.synchronize()
        return losses, {n: model.p(n).detach().cpu().numpy().copy() for n in model.names}, emb

    eager_losses, eager_params, eager_emb = run(False)
    graph_losses, graph_params, graph_emb = run(True)
    assert np.abs(eager_emb - graph_emb).max() < 5e-3, np.abs(eager_emb - graph_emb).max()
    assert np.allclose(eager_losses, graph_losses, rtol=1e-3, atol=1e-4), (eager_losses, graph_losses)
    assert eager_losses[-1] != eager_losses[0]
    # Parameters: AdamW moves a weight by ~lr per step whatever the size of its gradient, so a weight whose gradient is
    # rounding noise (the key bias - exactly zero in exact arithmetic - or a rarely hit row; the weight-gradient sums
    # are fp32 atomics, whose order differs from run to run) can end up 2 x 6 x lr apart between ANY two runs.  The gate
    # is therefore on the bulk: the two runs' parameters differ by a small fraction of how far training moved them.
    init = synthetic_state_dict(cfg)
    names = [n for n in eager_params if "key.bias" not in n]
    apart = sum(float(np.abs(eager_params[n] - graph_params[n]).sum()) for n in names)
    moved = sum(float(np.abs(eager_params[n] - init[n]).sum()) for n in names)
    print(f"graphed vs eager: parameters apart {apart:.4g} / moved {moved:.4g} = {apart / moved:.4f}; losses {graph_losses}")
    assert moved > 0 and apart / moved < 0.05, (apart, moved)


@pytest.mark.gpu
def test_fused_backward_refuses_hooks_and_frozen_parameters(gpu):
    """The HIP backward writes p.grad as a side effect (training._EncoderFunction.backward): uses that would silently
    get no gradient fail loudly instead - a hook on a parameter (DDP's mechanism), a single frozen
    parameter; the flat gradient buffer is what a data-parallel trainer reduces."""
    from semantic_search_kd_amd.bench_support import synthetic_ids
    from semantic_search_kd_amd.training import TrainableEncoder

    cfg = BertConfig(num_hidden_layers=1)
    dev = torch.device("cuda:0")
    model = TrainableEncoder(cfg, synthetic_state_dict(cfg), dev)
    ids, mask = synthetic_ids(2, 32, cfg.vocab_size, dev, seed=1)
    model(ids, mask).sum().backward()
    w = model.p("encoder.layer.0.output.dense.weight")
    assert w.grad is not None and w.grad.untyped_storage().data_ptr() == model.flat_grad.untyped_storage().data_ptr()
    assert float(model.flat_grad.abs().sum()) > 0
    handle = w.register_hook(lambda g: g)
    with pytest.raises(NotImplementedError, match="hooks"):
        model(ids, mask)
    handle.remove()
    model(ids, mask)
    w.requires_grad_(False)
    with pytest.raises(NotImplementedError, match="requires_grad"):
        model(ids, mask)


@pytest.mark.gpu
def test_two_branch_backward_equals_the_sum_of_its_halves(gpu):
    """sskd_generic_forward / _backward run batches of >= 2 x 16 384 tokens as two halves on two streams (csrc/train.hip
    generic_parts); the halves add into the SAME gradient buffers (atomic accumulation).  128 x 256 tokens split; the two
    64-row halves, called one after the other, do not.  Embeddings must agree bit for bit (rows do not interact), the
    gradient of the whole batch must equal the sum of the halves' gradients up to the order of fp32 additions."""
    from semantic_search_kd_amd.training import TrainableEncoder

    cfg = BertConfig(vocab_size=2000, num_hidden_layers=2)     # the e5-small-v2 widths: every weight gradient on gemm_tn384
    sd = synthetic_state_dict(cfg)
    B, S = 128, 256
    rng = np.random.default_rng(17)
    lengths = [int(x) for x in rng.integers(4, S + 1, size=B)]
    lengths[0] = lengths[64] = S
    ids, mask = enc_oracle.synthetic_token_ids(B, S, seed=41, vocab=2000, lengths=lengths)
    probe = torch.from_numpy(rng.standard_normal((B, cfg.hidden_size)).astype(np.float32)).cuda()

    def grads(rows):
        model = TrainableEncoder(cfg, sd, "cuda:0")
        embs = []
        for lo, hi in rows:
            emb = model(ids[lo:hi], mask[lo:hi], normalize=True)
            (emb * probe[lo:hi]).sum().backward()
            embs.append(emb.detach().clone())
        return torch.cat(embs), {n: model.p(n).grad.detach().clone() for n in model.names}

    e_whole, g_whole = grads([(0, B)])
    e_halves, g_halves = grads([(0, B // 2), (B // 2, B)])
    assert torch.equal(e_whole, e_halves)
    top = max(float(v.abs().max()) for v in g_halves.values())
    for name, ref in g_halves.items():
        got = g_whole[name]
        assert torch.isfinite(got).all(), name
        assert float((got - ref).abs().max()) <= 2e-3 * max(float(ref.abs().max()), 1e-3 * top), name
