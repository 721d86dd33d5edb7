#include "denoiser_torch.h"

#include <torch/csrc/jit/ir/constants.h>
#include <torch/csrc/jit/ir/ir.h>
#include <torch/csrc/jit/passes/inliner.h>
#include <torch/script.h>

#include "rto.h"

#include <iostream>
#include <stdexcept>

namespace rto {

struct TorchDenoiser::Impl {
    torch::jit::script::Module module;
    torch::Tensor weight, guidance;
    int device = 0;
    rto_guidance_net* fused = nullptr;  // non-null: forward runs librto's fused kernel
    int fused_levels = 0;
};

namespace {

// The reference's exporter traces a CLOSURE over the compact network (`torch.jit.trace(cast_and_forward, ...)`,
// denoiser/network.py:194-201): its ts_*.ts holds no parameters, the convolution weights are tensor
// constants of the graph.  Walk the (inlined) forward graph and collect {weight, bias} of every
// convolution node, in execution order.
std::vector<std::pair<torch::Tensor, torch::Tensor>> conv_constants(torch::jit::script::Module& m) {
    std::vector<std::pair<torch::Tensor, torch::Tensor>> out;
    std::shared_ptr<torch::jit::Graph> g = m.get_method("forward").graph()->copy();
    torch::jit::Inline(*g);
    for (torch::jit::Node* n : g->nodes()) {
        const std::string kind = n->kind().toQualString();
        if (kind != "aten::_convolution_mode" && kind != "aten::conv2d" && kind != "aten::_convolution" &&
            kind != "aten::convolution")
            continue;
        if (n->inputs().size() < 3) return {};
        const auto w = torch::jit::toIValue(n->input(1));
        const auto b = torch::jit::toIValue(n->input(2));
        if (!w || !w->isTensor() || !b || !b->isTensor()) return {};  // not constants: not the exporter's shape
        out.emplace_back(w->toTensor(), b->toTensor());
    }
    return out;
}

}  // namespace

TorchDenoiser::TorchDenoiser(const std::string& path, int device, bool use_fused) : impl_(new Impl) {
    if (path.empty()) throw std::runtime_error("No torchscript module is given to denoiser.");  // denoiser.cpp:13-16
    impl_->device = device;
    try {
        impl_->module = torch::jit::load(path, torch::Device(torch::kCUDA, (c10::DeviceIndex)device));
        impl_->module.eval();
    } catch (const c10::Error& e) {  // denoiser.cpp:22-26
        std::cerr << e.what() << std::endl;
        throw std::runtime_error("Error when loading torchscript model from " + path);
    }
    if (use_fused) {
        torch::Tensor w1, b1, w2, b2;
        int n_params = 0;
        for (const auto& p : impl_->module.named_parameters()) {
            ++n_params;
            if (p.name == "layers.0.conv.weight") w1 = p.value;
            if (p.name == "layers.0.conv.bias") b1 = p.value;
            if (p.name == "layers.1.conv.weight") w2 = p.value;
            if (p.name == "layers.1.conv.bias") b2 = p.value;
        }
        if (n_params == 0) {  // a reference-format file: weights are graph constants
            try {
                const auto convs = conv_constants(impl_->module);
                if (convs.size() == 2) {
                    w1 = convs[0].first;
                    b1 = convs[0].second;
                    w2 = convs[1].first;
                    b2 = convs[1].second;
                    n_params = 4;
                }
            } catch (const std::exception&) {  // unknown graph shape: stay on libtorch
                n_params = 0;
            }
        }
        if (n_params == 4 && w1.defined() && b1.defined() && w2.defined() && b2.defined() && w1.dim() == 4 &&
            w2.dim() == 4 && w1.size(1) == 8 && w1.size(2) == 3 && w1.size(3) == 3 && w2.size(1) == w1.size(0) &&
            w2.size(2) == 3 && w2.size(3) == 3) {
            auto host = [](const torch::Tensor& t) { return t.detach().to(torch::kCPU, torch::kFloat32).contiguous(); };
            const torch::Tensor hw1 = host(w1), hb1 = host(b1), hw2 = host(w2), hb2 = host(b2);
            rto_guidance_net* net = nullptr;
            const int c1 = (int)w1.size(0), levels = (int)w2.size(0) / 2;
            if (rto_guidance_net_create(hw1.data_ptr<float>(), hb1.data_ptr<float>(), hw2.data_ptr<float>(),
                                        hb2.data_ptr<float>(), c1, levels, device, &net) == RTO_OK) {
                impl_->fused = net;  // (RTO_E_UNSUPPORTED for other widths: stay on libtorch)
                impl_->fused_levels = levels;
                // Recognising a module by its tensors says nothing about the rest of its graph: run both
                // routes once on a small random input and keep the fused kernel only if they agree to fp16
                // convolution accuracy (|d guidance| on a [0, 6] scale, |d weight| on [0, 1]).
                try {
                    torch::NoGradGuard no_grad;
                    const auto opts = torch::TensorOptions().device(torch::kCUDA, device).dtype(torch::kFloat32);
                    torch::manual_seed(20230418);
                    torch::Tensor aux = torch::rand({1, 8, 24, 40}, opts);
                    auto maps = impl_->module.forward({aux}).toTuple()->elements();
                    torch::Tensor wm = torch::empty({1, levels, 24, 40}, opts), gm = torch::empty({1, levels, 24, 40}, opts);
                    const bool ran = rto_guidance_net_forward(net, nullptr, aux.data_ptr<float>(), 1, 24, 40,
                                                              wm.data_ptr<float>(), gm.data_ptr<float>()) == RTO_OK;
                    const double dw = ran ? (wm - maps[0].toTensor().to(torch::kFloat32)).abs().max().item<double>() : 1e9;
                    const double dg = ran ? (gm - maps[1].toTensor().to(torch::kFloat32)).abs().max().item<double>() : 1e9;
                    if (!(dw < 2e-2) || !(dg < 5e-2)) {
                        std::cerr << "INFO: the ts module is not the compact GuidanceNet the fused kernel implements (|dw| " << dw
                                  << ", |dg| " << dg << "): using libtorch" << std::endl;
                        rto_guidance_net_free(net);
                        impl_->fused = nullptr;
                    }
                } catch (const std::exception& e) {
                    rto_guidance_net_free(net);
                    impl_->fused = nullptr;
                }
            }
        }
    }
}

TorchDenoiser::~TorchDenoiser() {
    if (impl_ && impl_->fused) rto_guidance_net_free(impl_->fused);
}

bool TorchDenoiser::fused() const { return impl_->fused != nullptr; }
rto_guidance_net* TorchDenoiser::fused_handle() const { return impl_->fused; }

void TorchDenoiser::forward(float* aux, int n, int H, int W, const float** weight, const float** guidance, int* levels, bool input_rgba) {
    torch::NoGradGuard no_grad;
    const auto options = torch::TensorOptions().device(torch::kCUDA, impl_->device).dtype(torch::kFloat32);
    if (impl_->fused) {  // same maps from one HIP kernel on the default stream (the CLI's stream)
        const int L = impl_->fused_levels;
        if (!impl_->weight.defined() || impl_->weight.size(0) != n || impl_->weight.size(2) != H || impl_->weight.size(3) != W) {
            impl_->weight = torch::empty({n, L, H, W}, options);
            impl_->guidance = torch::empty({n, L, H, W}, options);
        }
        // aux is the renderer's buffer: planes 4..7 are the squares of planes 0..3
        if (rto_guidance_net_forward_ex(impl_->fused, nullptr, aux, n, H, W, impl_->weight.data_ptr<float>(),
                                        impl_->guidance.data_ptr<float>(), input_rgba ? RTO_NET_INPUT_RGBA : RTO_NET_AUX_SQUARES_IMPLIED) != RTO_OK)
            throw std::runtime_error(std::string("fused GuidanceNet failed: ") + rto_last_error());
        *weight = impl_->weight.data_ptr<float>();
        *guidance = impl_->guidance.data_ptr<float>();
        *levels = L;
        return;
    }
    if (input_rgba) throw std::runtime_error("the TorchScript module reads the 8-plane aux buffer: lean outputs need the fused network");
    torch::Tensor aux_t = torch::from_blob(aux, {n, 8, H, W}, options);  // denoiser.cpp:40-43 (n = 1 there)
    auto maps = impl_->module.forward({aux_t}).toTuple()->elements();
    impl_->weight = maps[0].toTensor().contiguous();    // [n,L,H,W]
    impl_->guidance = maps[1].toTensor().contiguous();  // [n,L,H,W]
    *weight = impl_->weight.data_ptr<float>();
    *guidance = impl_->guidance.data_ptr<float>();
    *levels = (int)impl_->guidance.size(1);
}

}  // namespace rto
