// libdisenlink_torch.so — the COMPILED torch binding over the C ABI of libdisenlink_hip.so (include/disenlink_hip.h):
// BASELINE.json's north star asks for "a PyTorch-ROCm extension exposing a thin C-ABI".  The C ABI is the product
// boundary; disenlink_amd/ops.py reaches it through ctypes from Python autograd.Functions, and this file reaches the SAME
// entry points from a C++ autograd node registered with TORCH_LIBRARY — no Python frame between the launches of a
// training step's hot path (route, aggregate, one-pass scorer, loss value; backward: routing / aggregation), which is
// what the eager loop on small graphs spends its host time in.
//
//   torch.ops.disenlink_native.hot_path_pairs_loss(Z, graph_ptr, inc_ptr, n_edges, beta, t, label, weight,
//                                                  ws_graph, ws_pairs, ws_bce) -> (H, prob, loss)
//
// graph_ptr / inc_ptr: addresses of the dl_graph / dl_pair_incidence structs the Python Graph / PairList objects own
// (they must outlive the call and its backward: disenlink_amd/native.py hangs the Python objects on the node's metadata);
// ws_*: the caller's scratch tensors (the C ABI never allocates).
// Replaces model.py:56-75, 109-113 + the loss of main_disentangled.py:195 on a pair list, like ops.HotPathPairsLoss —
// same kernels, same bits, also for a gradient arriving on `prob` or on the embedding.  fp32 tables.
// Built by disenlink_amd/build.py with g++ against the installed torch (no device code in this file).
#include <torch/library.h>
#include <torch/autograd.h>
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include "disenlink_hip.h"

// Operator set + schemas of this binding; native.py refuses a library built for another number (a stale .so would
// otherwise fail deep inside a training step with an AttributeError or a schema error).  Bump with every schema change.
#define DL_TORCH_BINDING_ABI 6

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void check(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed (", rc, "): ", dl_last_error());
}

void* stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

struct HotPathPairsLoss : public torch::autograd::Function<HotPathPairsLoss> {
    static variable_list forward(AutogradContext* ctx, const Tensor& Z_in, int64_t graph_ptr, int64_t inc_ptr, int64_t n_edges,
                                 double beta, double t, const Tensor& label_in, const Tensor& weight_in, const Tensor& ws_g,
                                 const Tensor& ws_p, const Tensor& ws_b, int64_t table_bf16) {
        TORCH_CHECK(Z_in.is_cuda() && Z_in.dim() == 3 && Z_in.scalar_type() == at::kFloat, "Z must be a CUDA fp32 [N,K,d] tensor");
        TORCH_CHECK(label_in.is_cuda() && weight_in.is_cuda() && label_in.numel() == weight_in.numel(), "label / weight");
        at::AutoDispatchBelowADInplaceOrView guard;
        // table_bf16: the gathered Z / H tables are stored as bf16 (arithmetic, per-edge values and all gradients stay fp32)
        const dl_dtype dt = table_bf16 ? DL_BF16 : DL_F32;
        const Tensor Z = table_bf16 ? Z_in.contiguous().to(at::kBFloat16) : Z_in.contiguous();
        const Tensor label = label_in.to(at::kFloat).contiguous(), weight = weight_in.to(at::kFloat).contiguous();
        const auto* g = reinterpret_cast<const dl_graph*>(graph_ptr);
        const auto* inc = reinterpret_cast<const dl_pair_incidence*>(inc_ptr);
        const int64_t N = Z.size(0);
        const int K = (int)Z.size(1), d = (int)Z.size(2);
        const int64_t P = label.numel();
        const auto f32 = Z.options().dtype(at::kFloat), u8 = Z.options().dtype(at::kByte);
        Tensor p = at::empty({n_edges}, u8), a = at::empty({n_edges}, f32), s = at::empty({N, K}, f32);
        Tensor H = at::empty_like(Z), prob = at::empty({P}, f32), dZs = at::empty(Z.sizes(), f32), dHs = at::empty(Z.sizes(), f32);
        Tensor loss = at::empty({1}, f32), gbce = at::empty({P}, f32);
        void* st = stream();
        check(dl_route_fwd(g, Z.data_ptr(), K, d, dt, (float)t, p.data_ptr<uint8_t>(), a.data_ptr<float>(), s.data_ptr<float>(),
                           ws_g.data_ptr(), (size_t)ws_g.numel(), st), "dl_route_fwd");
        check(dl_aggregate_fwd(g, Z.data_ptr(), K, d, dt, (float)beta, p.data_ptr<uint8_t>(), a.data_ptr<float>(),
                               s.data_ptr<float>(), H.data_ptr(), ws_g.data_ptr(), (size_t)ws_g.numel(), st), "dl_aggregate_fwd");
        check(dl_score_pairs_train(Z.data_ptr(), H.data_ptr(), K, d, dt, (float)t, inc, label.data_ptr<float>(),
                                   weight.data_ptr<float>(), prob.data_ptr<float>(), dZs.data_ptr<float>(), dHs.data_ptr<float>(),
                                   ws_p.data_ptr(), (size_t)ws_p.numel(), st), "dl_score_pairs_train");
        check(dl_pair_bce(prob.data_ptr<float>(), label.data_ptr<float>(), weight.data_ptr<float>(), (int)P, loss.data_ptr<float>(),
                          gbce.data_ptr<float>(), ws_b.data_ptr(), (size_t)ws_b.numel(), st), "dl_pair_bce");
        ctx->saved_data["graph"] = graph_ptr;
        ctx->saved_data["inc"] = inc_ptr;
        ctx->saved_data["beta"] = beta;
        ctx->saved_data["t"] = t;
        ctx->saved_data["bf16"] = table_bf16;
        // H and prob are outputs of this node (autograd handles saved outputs without a reference cycle); they and the
        // pair workspace serve the general backward (a gradient arriving on prob)
        ctx->save_for_backward({Z, p, a, s, dZs, dHs, ws_g, H, prob, ws_p});       // (bf16: H here is the bf16 table, not the output)
        ctx->set_materialize_grads(false);
        Tensor loss0 = loss.select(0, 0);
        return {table_bf16 ? H.to(at::kFloat) : H, prob, loss0};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor &Z = saved[0], &p = saved[1], &a = saved[2], &s = saved[3], &dZs = saved[4], &dHs = saved[5], &ws_g = saved[6];
        const Tensor &H = saved[7], &prob = saved[8], &ws_p = saved[9];
        const auto* g = reinterpret_cast<const dl_graph*>(ctx->saved_data["graph"].toInt());
        const auto* inc = reinterpret_cast<const dl_pair_incidence*>(ctx->saved_data["inc"].toInt());
        const float beta = (float)ctx->saved_data["beta"].toDouble(), t = (float)ctx->saved_data["t"].toDouble();
        const int K = (int)Z.size(1), d = (int)Z.size(2);
        const dl_dtype dt = ctx->saved_data["bf16"].toInt() ? DL_BF16 : DL_F32;
        const Tensor& g_emb = grads[0];
        const Tensor& g_prob = grads[1];
        const Tensor& g_loss = grads[2];
        at::AutoDispatchBelowADInplaceOrView guard;
        void* st = stream();
        Tensor dZ;
        if (g_loss.defined() && !g_emb.defined() && !g_prob.defined()) {
            // loss.backward(): everything downstream is linear in the scorer's gradients, so d/dloss scales the RESULT
            // inside the last kernel (dl_route_aggregate_bwd_scaled) — no scaling passes over the two [N,K,d] arrays
            dZ = at::empty(Z.sizes(), dZs.options());
            const Tensor scale = g_loss.to(at::kFloat).reshape({1}).contiguous();
            check(dl_route_aggregate_bwd_scaled(g, Z.data_ptr(), K, d, dt, beta, t, p.data_ptr<uint8_t>(), a.data_ptr<float>(),
                                                s.data_ptr<float>(), dHs.data_ptr<float>(), dZs.data_ptr<float>(),
                                                scale.data_ptr<float>(), dZ.data_ptr<float>(), ws_g.data_ptr(),
                                                (size_t)ws_g.numel(), st), "dl_route_aggregate_bwd_scaled");
        } else {
            // the general case, in the order of ops.HotPathPairsLoss.backward (same bits): the loss's share, then another
            // function of the scores through the ordinary scorer backward (dl_score_pairs_bwd, recompute form), then a
            // gradient on the embedding
            Tensor dH = g_loss.defined() ? dHs * g_loss : at::zeros_like(dHs);
            dZ = g_loss.defined() ? dZs * g_loss : at::zeros_like(dZs);
            if (g_prob.defined()) {
                const Tensor gp = g_prob.to(at::kFloat).contiguous();
                Tensor dZ2 = at::empty(Z.sizes(), dZs.options()), dH2 = at::empty(Z.sizes(), dZs.options());
                check(dl_score_pairs_bwd(Z.data_ptr(), H.data_ptr(), K, d, dt, t, inc, prob.data_ptr<float>(), gp.data_ptr<float>(),
                                         nullptr, dZ2.data_ptr<float>(), dH2.data_ptr<float>(), ws_p.data_ptr(), (size_t)ws_p.numel(),
                                         st), "dl_score_pairs_bwd");
                dZ.add_(dZ2);
                dH.add_(dH2);
            }
            if (g_emb.defined()) dH.add_(g_emb.to(at::kFloat).reshape(dH.sizes()));
            dH = dH.contiguous();
            check(dl_route_aggregate_bwd(g, Z.data_ptr(), K, d, dt, beta, t, p.data_ptr<uint8_t>(), a.data_ptr<float>(),
                                         s.data_ptr<float>(), dH.data_ptr<float>(), dZ.data_ptr<float>(), 1, ws_g.data_ptr(),
                                         (size_t)ws_g.numel(), st), "dl_route_aggregate_bwd");
        }
        return {dZ, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

std::tuple<Tensor, Tensor, Tensor> hot_path_pairs_loss(const Tensor& Z, int64_t graph_ptr, int64_t inc_ptr, int64_t n_edges,
                                                       double beta, double t, const Tensor& label, const Tensor& weight,
                                                       const Tensor& ws_g, const Tensor& ws_p, const Tensor& ws_b, int64_t table_bf16) {
    auto out = HotPathPairsLoss::apply(Z, graph_ptr, inc_ptr, n_edges, beta, t, label, weight, ws_g, ws_p, ws_b, table_bf16);
    return {out[0], out[1], out[2]};
}

// ---------------------------------------------------------------------------- projection over the module's shared buffers
// ops.ProjectStacked in C++ (model.py:24-27, 106 and their autograd, main_disentangled.py:198): the K per-factor Parameters
// are the autograd inputs (`params`: K x mlp1.weight, K x mlp1.bias, K x mlp2.weight, K x mlp2.bias — views of the four
// stacked buffers W1 [K,nhid,F], b1 [K,nhid], W2 [K,d,nhid], b2 [K,d] the kernels read), every parameter's gradient is a
// slice of ONE flat allocation [dW1 | db1 | dW2 | db2] (the optimiser and the sharded all-reduce take it as it is).
// Two-layer form, F % 4 == 0, d in {32, 64, 128}; everything else stays with the Python operator.
struct ProjectStackedNode : public torch::autograd::Function<ProjectStackedNode> {
    // (params as at::TensorList: a std::vector<Tensor> argument is NOT seen as a list of autograd inputs by Function::apply —
    // the output then carries no grad_fn)
    static Tensor forward(AutogradContext* ctx, at::TensorList params, const Tensor& x_in, const Tensor& W1, const Tensor& b1,
                          const Tensor& W2, const Tensor& b2, bool keep_hid, const c10::optional<Tensor>& xplanes) {
        TORCH_CHECK(x_in.is_cuda() && x_in.scalar_type() == at::kFloat && x_in.dim() == 2, "x must be a CUDA fp32 [N,F] tensor");
        TORCH_CHECK(W1.is_contiguous() && b1.is_contiguous() && W2.is_contiguous() && b2.is_contiguous(), "stacked buffers must be dense");
        at::AutoDispatchBelowADInplaceOrView guard;
        const Tensor x = x_in.contiguous();
        const int N = (int)x.size(0), F = (int)x.size(1), K = (int)W1.size(0), nhid = (int)W1.size(1), d = (int)W2.size(1);
        TORCH_CHECK(W1.size(2) == F && F % 4 == 0 && dl_project_supported(d), "shape not served by the compiled projection node");
        TORCH_CHECK((int64_t)params.size() == 4 * (int64_t)K, "params must be the 4 K per-factor parameters");
        Tensor Z = at::empty({N, K, d}, x.options());
        Tensor hid;
        if (keep_hid) hid = at::empty({(int64_t)dl_project_hidden_floats(N, K, nhid)}, x.options());
        Tensor ws = at::empty({(int64_t)dl_project_fwd_workspace_bytes(N, F, K, nhid, d, 1) + 256}, x.options().dtype(at::kByte));
        // xplanes: the persistent bf16 planes of x and x^T (dl_project_xplanes_build; ops._XPlanes) — valid for this x only
        const bool have_xp = xplanes.has_value() && xplanes->defined() && x.data_ptr() == x_in.data_ptr();
        check(dl_project_fwd_xp(x.data_ptr<float>(), N, F, K, nhid, d, W1.data_ptr<float>(), b1.data_ptr<float>(), W2.data_ptr<float>(),
                                b2.data_ptr<float>(), Z.data_ptr<float>(), keep_hid ? hid.data_ptr<float>() : nullptr, ws.data_ptr(),
                                (size_t)ws.numel(), have_xp ? xplanes->data_ptr() : nullptr, stream()), "dl_project_fwd");
        ctx->save_for_backward({x, W1, b1, W2, keep_hid ? hid : Tensor(), have_xp ? *xplanes : Tensor()});
        ctx->saved_data["K"] = (int64_t)K;
        return Z;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor &x = saved[0], &W1 = saved[1], &b1 = saved[2], &W2 = saved[3], &hid = saved[4], &xp = saved[5];
        const int64_t K = ctx->saved_data["K"].toInt();
        const int N = (int)x.size(0), F = (int)x.size(1), nhid = (int)W1.size(1), d = (int)W2.size(1);
        at::AutoDispatchBelowADInplaceOrView guard;
        const Tensor dZ = grads[0].to(at::kFloat).contiguous();
        const int64_t n1 = W1.numel(), n2 = b1.numel(), n3 = W2.numel(), n4 = K * d;
        Tensor flat = at::empty({n1 + n2 + n3 + n4}, x.options());
        Tensor dW1 = flat.narrow(0, 0, n1).view(W1.sizes()), db1 = flat.narrow(0, n1, n2).view(b1.sizes());
        Tensor dW2 = flat.narrow(0, n1 + n2, n3).view(W2.sizes()), db2 = flat.narrow(0, n1 + n2 + n3, n4).view({K, d});
        Tensor ws = at::empty({(int64_t)dl_project_bwd_workspace_bytes(N, F, (int)K, nhid, d, 1) + 256}, x.options().dtype(at::kByte));
        check(dl_project_bwd_xp(x.data_ptr<float>(), N, F, (int)K, nhid, d, W1.data_ptr<float>(), b1.data_ptr<float>(), W2.data_ptr<float>(),
                                dZ.data_ptr<float>(), hid.defined() ? hid.data_ptr<float>() : nullptr, dW1.data_ptr<float>(),
                                db1.data_ptr<float>(), dW2.data_ptr<float>(), db2.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(),
                                xp.defined() ? xp.data_ptr() : nullptr, stream()), "dl_project_bwd");
        variable_list out;
        out.reserve(4 * K + 7);
        for (const Tensor* g : {&dW1, &db1, &dW2, &db2})
            for (int64_t k = 0; k < K; ++k) out.push_back(g->select(0, k));
        for (int i = 0; i < 7; ++i) out.push_back(Tensor());                      // x, W1, b1, W2, b2, keep_hid, xplanes
        return out;
    }
};

Tensor project_stacked(const Tensor& x, const Tensor& W1, const Tensor& b1, const Tensor& W2, const Tensor& b2,
                       at::TensorList params, bool keep_hid, const c10::optional<Tensor>& xplanes) {
    return ProjectStackedNode::apply(params, x, W1, b1, W2, b2, keep_hid, xplanes);
}

// ---------------------------------------------------------------------------- Adam over the shared buffers (main_disentangled.py:150, 199)
// optim.StackedAdam.step without Python between the gradient bookkeeping and the launch: the K gradients of a group are the
// slices of one stacked tensor when the projection's backward made them (checked by address); otherwise they are stacked.
// host_step > 0: the caller counts the steps (dl_adam_step_at: one launch instead of two); 0: the counter in `state` (graph capture).
void adam_step(at::TensorList bufs, at::TensorList params, at::TensorList exp_avg, at::TensorList exp_avg_sq, const Tensor& state,
               double lr, double beta1, double beta2, double eps, double weight_decay, int64_t host_step) {
    const int n = (int)bufs.size();
    TORCH_CHECK(n >= 1 && n <= DL_ADAM_MAX_BUFS && (int)exp_avg.size() == n && (int)exp_avg_sq.size() == n, "adam_step: buffer lists");
    TORCH_CHECK(params.size() % n == 0, "adam_step: params must hold the same number of parameters per buffer");
    const int64_t K = (int64_t)params.size() / n;
    std::vector<Tensor> keep;                                                       // stacked copies stay alive until the launch is queued
    float* pp[DL_ADAM_MAX_BUFS]; const float* gp[DL_ADAM_MAX_BUFS]; float* mp[DL_ADAM_MAX_BUFS]; float* vp[DL_ADAM_MAX_BUFS];
    size_t numel[DL_ADAM_MAX_BUFS];
    for (int b = 0; b < n; ++b) {
        const Tensor& g0 = params[b * K].grad();
        TORCH_CHECK(g0.defined(), "adam_step: a parameter has no gradient");
        bool adjacent = g0.is_contiguous() && g0.scalar_type() == at::kFloat && g0.numel() * K == bufs[b].numel();
        const char* base = adjacent ? static_cast<const char*>(g0.data_ptr()) : nullptr;
        const size_t step = (size_t)g0.numel() * sizeof(float);
        for (int64_t k = 1; adjacent && k < K; ++k) {
            const Tensor& gk = params[b * K + k].grad();
            adjacent = gk.defined() && gk.is_contiguous() && static_cast<const char*>(gk.data_ptr()) == base + k * step;
        }
        if (adjacent) {
            const auto& st = g0.storage();
            adjacent = (size_t)st.nbytes() >= (size_t)(base - static_cast<const char*>(st.data())) + (size_t)K * step;
        }
        if (adjacent) {
            gp[b] = g0.data_ptr<float>();
        } else {
            std::vector<Tensor> gs;
            for (int64_t k = 0; k < K; ++k) {
                TORCH_CHECK(params[b * K + k].grad().defined(), "adam_step: parameter ", b * K + k, " (buffer ", b, ", factor ", k,
                            ") has no gradient — every factor of a stacked buffer must have taken part in the backward pass");
                gs.push_back(params[b * K + k].grad().to(at::kFloat));
            }
            keep.push_back(at::stack(gs).contiguous());
            gp[b] = keep.back().data_ptr<float>();
        }
        pp[b] = bufs[b].data_ptr<float>();
        mp[b] = exp_avg[b].data_ptr<float>();
        vp[b] = exp_avg_sq[b].data_ptr<float>();
        numel[b] = (size_t)bufs[b].numel();
    }
    if (host_step > 0)
        check(dl_adam_step_at(n, pp, gp, mp, vp, numel, state.data_ptr<float>(), (long long)host_step, lr, beta1, beta2, eps, weight_decay,
                              stream()), "dl_adam_step_at");
    else
        check(dl_adam_step(n, pp, gp, mp, vp, numel, state.data_ptr<float>(), lr, beta1, beta2, eps, weight_decay, stream()), "dl_adam_step");
}

// ---------------------------------------------------------------------------- tie-averaged AUC counts (main_disentangled.py:202-204)
Tensor auc_pair_counts(const Tensor& score, const Tensor& pos_idx, const Tensor& neg_idx) {
    TORCH_CHECK(score.is_cuda() && score.scalar_type() == at::kFloat && score.is_contiguous(), "score must be a dense CUDA fp32 vector");
    TORCH_CHECK(pos_idx.scalar_type() == at::kLong && neg_idx.scalar_type() == at::kLong, "index sets must be int64");
    Tensor u2 = at::empty({1}, score.options().dtype(at::kLong));
    check(dl_auc_pair_counts(score.data_ptr<float>(), pos_idx.data_ptr<int64_t>(), (int)pos_idx.numel(), neg_idx.data_ptr<int64_t>(),
                             (int)neg_idx.numel(), reinterpret_cast<unsigned long long*>(u2.data_ptr<int64_t>()), stream()),
          "dl_auc_pair_counts");
    return u2;
}

// ---------------------------------------------------------------------------- end of an epoch on the device (main_disentangled.py:199-214)
// early_stop.DeviceEarlyStop.finish without ctypes: the validation counts of `score_val` added to u2, then dl_epoch_finish
// (AUC, best-weights copy, patience, history; ring_ptr = address of the pinned host ring, 0 for none).
void epoch_finish(const Tensor& score_val, const Tensor& pos_idx, const Tensor& neg_idx, const Tensor& u2, const Tensor& loss,
                  at::TensorList params, at::TensorList best, const Tensor& state, const Tensor& hist, int64_t ring_ptr,
                  int64_t ring, double denom2, int64_t max_epochs, int64_t patience) {
    TORCH_CHECK(score_val.is_cuda() && score_val.scalar_type() == at::kFloat && score_val.is_contiguous(), "score_val: dense CUDA fp32");
    TORCH_CHECK(loss.is_cuda() && loss.scalar_type() == at::kFloat && loss.numel() == 1, "loss: one CUDA fp32 value");
    TORCH_CHECK(pos_idx.scalar_type() == at::kLong && neg_idx.scalar_type() == at::kLong && u2.scalar_type() == at::kLong, "index sets / u2: int64");
    const int n = (int)params.size();
    TORCH_CHECK(n <= DL_ADAM_MAX_BUFS && (int)best.size() == n, "epoch_finish: buffer lists");
    TORCH_CHECK((size_t)state.numel() * state.element_size() >= dl_epoch_state_bytes() && hist.scalar_type() == at::kDouble, "state / hist");
    const float* pp[DL_ADAM_MAX_BUFS]; float* bp[DL_ADAM_MAX_BUFS]; size_t numel[DL_ADAM_MAX_BUFS];
    for (int b = 0; b < n; ++b) {
        TORCH_CHECK(params[b].scalar_type() == at::kFloat && params[b].is_contiguous() && best[b].is_contiguous() &&
                    best[b].numel() == params[b].numel(), "epoch_finish: buffer ", b);
        pp[b] = params[b].data_ptr<float>(); bp[b] = best[b].data_ptr<float>(); numel[b] = (size_t)params[b].numel();
    }
    auto* u2p = reinterpret_cast<unsigned long long*>(u2.data_ptr<int64_t>());
    check(dl_auc_pair_counts_add(score_val.data_ptr<float>(), pos_idx.data_ptr<int64_t>(), (int)pos_idx.numel(),
                                 neg_idx.data_ptr<int64_t>(), (int)neg_idx.numel(), u2p, stream()), "dl_auc_pair_counts_add");
    check(dl_epoch_finish(n, pp, bp, numel, loss.data_ptr<float>(), u2p, denom2, state.data_ptr(), hist.data_ptr<double>(),
                          (long long)max_epochs, (long long)patience, reinterpret_cast<double*>(ring_ptr), (int)ring, stream()),
          "dl_epoch_finish");
}

}  // namespace

TORCH_LIBRARY(disenlink_native, m) {
    m.def("hot_path_pairs_loss(Tensor Z, int graph_ptr, int inc_ptr, int n_edges, float beta, float t, Tensor label, "
          "Tensor weight, Tensor ws_graph, Tensor ws_pairs, Tensor ws_bce, int table_bf16) -> (Tensor, Tensor, Tensor)");
    m.def("project_stacked(Tensor x, Tensor W1, Tensor b1, Tensor W2, Tensor b2, Tensor[] params, bool keep_hid, Tensor? xplanes) -> Tensor");
    m.def("adam_step(Tensor[] bufs, Tensor[] params, Tensor[] exp_avg, Tensor[] exp_avg_sq, Tensor state, float lr, float beta1, "
          "float beta2, float eps, float weight_decay, int host_step=0) -> ()");
    m.def("auc_pair_counts(Tensor score, Tensor pos_idx, Tensor neg_idx) -> Tensor");
    m.def("epoch_finish(Tensor score_val, Tensor pos_idx, Tensor neg_idx, Tensor u2, Tensor loss, Tensor[] params, Tensor[] best, "
          "Tensor state, Tensor hist, int ring_ptr, int ring, float denom2, int max_epochs, int patience) -> ()");
    m.def("abi_version() -> str");
    m.def("binding_abi() -> int");
}

TORCH_LIBRARY_IMPL(disenlink_native, Autograd, m) {
    m.impl("hot_path_pairs_loss", hot_path_pairs_loss);
    m.impl("project_stacked", project_stacked);
}
TORCH_LIBRARY_IMPL(disenlink_native, CUDA, m) {
    m.impl("hot_path_pairs_loss", hot_path_pairs_loss);      // (no-grad calls)
    m.impl("project_stacked", project_stacked);
    m.impl("adam_step", adam_step);
    m.impl("auc_pair_counts", auc_pair_counts);
    m.impl("epoch_finish", epoch_finish);
}
TORCH_LIBRARY_IMPL(disenlink_native, CompositeExplicitAutograd, m) {
    m.impl("abi_version", []() { return std::string(dl_version()); });
    m.impl("binding_abi", []() { return (int64_t)DL_TORCH_BINDING_ABI; });
}
