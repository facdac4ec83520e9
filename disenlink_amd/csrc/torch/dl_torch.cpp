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
                                 const Tensor& ws_p, const Tensor& ws_b) {
        TORCH_CHECK(Z_in.is_cuda() && Z_in.dim() == 3 && Z_in.scalar_type() == at::kFloat, "Z must be a CUDA fp32 [N,K,d] tensor");
        TORCH_CHECK(label_in.is_cuda() && weight_in.is_cuda() && label_in.numel() == weight_in.numel(), "label / weight");
        at::AutoDispatchBelowADInplaceOrView guard;
        const Tensor Z = Z_in.contiguous(), label = label_in.to(at::kFloat).contiguous(), weight = weight_in.to(at::kFloat).contiguous();
        const auto* g = reinterpret_cast<const dl_graph*>(graph_ptr);
        const auto* inc = reinterpret_cast<const dl_pair_incidence*>(inc_ptr);
        const int64_t N = Z.size(0);
        const int K = (int)Z.size(1), d = (int)Z.size(2);
        const int64_t P = label.numel();
        const auto f32 = Z.options(), u8 = Z.options().dtype(at::kByte);
        Tensor p = at::empty({n_edges}, u8), a = at::empty({n_edges}, f32), s = at::empty({N, K}, f32);
        Tensor H = at::empty_like(Z), prob = at::empty({P}, f32), dZs = at::empty_like(Z), dHs = at::empty_like(Z);
        Tensor loss = at::empty({1}, f32), gbce = at::empty({P}, f32);
        void* st = stream();
        check(dl_route_fwd(g, Z.data_ptr(), K, d, DL_F32, (float)t, p.data_ptr<uint8_t>(), a.data_ptr<float>(), s.data_ptr<float>(),
                           ws_g.data_ptr(), (size_t)ws_g.numel(), st), "dl_route_fwd");
        check(dl_aggregate_fwd(g, Z.data_ptr(), K, d, DL_F32, (float)beta, p.data_ptr<uint8_t>(), a.data_ptr<float>(),
                               s.data_ptr<float>(), H.data_ptr(), ws_g.data_ptr(), (size_t)ws_g.numel(), st), "dl_aggregate_fwd");
        check(dl_score_pairs_train(Z.data_ptr(), H.data_ptr(), K, d, DL_F32, (float)t, inc, label.data_ptr<float>(),
                                   weight.data_ptr<float>(), prob.data_ptr<float>(), dZs.data_ptr<float>(), dHs.data_ptr<float>(),
                                   ws_p.data_ptr(), (size_t)ws_p.numel(), st), "dl_score_pairs_train");
        check(dl_pair_bce(prob.data_ptr<float>(), label.data_ptr<float>(), weight.data_ptr<float>(), (int)P, loss.data_ptr<float>(),
                          gbce.data_ptr<float>(), ws_b.data_ptr(), (size_t)ws_b.numel(), st), "dl_pair_bce");
        ctx->saved_data["graph"] = graph_ptr;
        ctx->saved_data["inc"] = inc_ptr;
        ctx->saved_data["beta"] = beta;
        ctx->saved_data["t"] = t;
        // H and prob are outputs of this node (autograd handles saved outputs without a reference cycle); they and the
        // pair workspace serve the general backward (a gradient arriving on prob)
        ctx->save_for_backward({Z, p, a, s, dZs, dHs, ws_g, H, prob, ws_p});
        ctx->set_materialize_grads(false);
        Tensor loss0 = loss.select(0, 0);
        return {H, prob, loss0};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor &Z = saved[0], &p = saved[1], &a = saved[2], &s = saved[3], &dZs = saved[4], &dHs = saved[5], &ws_g = saved[6];
        const Tensor &H = saved[7], &prob = saved[8], &ws_p = saved[9];
        const auto* g = reinterpret_cast<const dl_graph*>(ctx->saved_data["graph"].toInt());
        const auto* inc = reinterpret_cast<const dl_pair_incidence*>(ctx->saved_data["inc"].toInt());
        const float beta = (float)ctx->saved_data["beta"].toDouble(), t = (float)ctx->saved_data["t"].toDouble();
        const int K = (int)Z.size(1), d = (int)Z.size(2);
        const Tensor& g_emb = grads[0];
        const Tensor& g_prob = grads[1];
        const Tensor& g_loss = grads[2];
        at::AutoDispatchBelowADInplaceOrView guard;
        void* st = stream();
        Tensor dZ;
        if (g_loss.defined() && !g_emb.defined() && !g_prob.defined()) {
            // loss.backward(): everything downstream is linear in the scorer's gradients, so d/dloss scales the RESULT
            // inside the last kernel (dl_route_aggregate_bwd_scaled) — no scaling passes over the two [N,K,d] arrays
            dZ = at::empty_like(Z);
            const Tensor scale = g_loss.to(at::kFloat).reshape({1}).contiguous();
            check(dl_route_aggregate_bwd_scaled(g, Z.data_ptr(), K, d, DL_F32, beta, t, p.data_ptr<uint8_t>(), a.data_ptr<float>(),
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
                Tensor dZ2 = at::empty_like(Z), dH2 = at::empty_like(Z);
                check(dl_score_pairs_bwd(Z.data_ptr(), H.data_ptr(), K, d, DL_F32, t, inc, prob.data_ptr<float>(), gp.data_ptr<float>(),
                                         nullptr, dZ2.data_ptr<float>(), dH2.data_ptr<float>(), ws_p.data_ptr(), (size_t)ws_p.numel(),
                                         st), "dl_score_pairs_bwd");
                dZ.add_(dZ2);
                dH.add_(dH2);
            }
            if (g_emb.defined()) dH.add_(g_emb.to(at::kFloat).reshape(dH.sizes()));
            dH = dH.contiguous();
            check(dl_route_aggregate_bwd(g, Z.data_ptr(), K, d, DL_F32, beta, t, p.data_ptr<uint8_t>(), a.data_ptr<float>(),
                                         s.data_ptr<float>(), dH.data_ptr<float>(), dZ.data_ptr<float>(), 1, ws_g.data_ptr(),
                                         (size_t)ws_g.numel(), st), "dl_route_aggregate_bwd");
        }
        return {dZ, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

std::tuple<Tensor, Tensor, Tensor> hot_path_pairs_loss(const Tensor& Z, int64_t graph_ptr, int64_t inc_ptr, int64_t n_edges,
                                                       double beta, double t, const Tensor& label, const Tensor& weight,
                                                       const Tensor& ws_g, const Tensor& ws_p, const Tensor& ws_b) {
    auto out = HotPathPairsLoss::apply(Z, graph_ptr, inc_ptr, n_edges, beta, t, label, weight, ws_g, ws_p, ws_b);
    return {out[0], out[1], out[2]};
}

}  // namespace

TORCH_LIBRARY(disenlink_native, m) {
    m.def("hot_path_pairs_loss(Tensor Z, int graph_ptr, int inc_ptr, int n_edges, float beta, float t, Tensor label, "
          "Tensor weight, Tensor ws_graph, Tensor ws_pairs, Tensor ws_bce) -> (Tensor, Tensor, Tensor)");
    m.def("abi_version() -> str");
}

TORCH_LIBRARY_IMPL(disenlink_native, Autograd, m) { m.impl("hot_path_pairs_loss", hot_path_pairs_loss); }
TORCH_LIBRARY_IMPL(disenlink_native, CUDA, m) { m.impl("hot_path_pairs_loss", hot_path_pairs_loss); }      // (no-grad calls)
TORCH_LIBRARY_IMPL(disenlink_native, CompositeExplicitAutograd, m) {
    m.impl("abi_version", []() { return std::string(dl_version()); });
}
