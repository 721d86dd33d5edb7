// volrend_headless -- offscreen PlenOctree rendering with batched regular tracking + GuidanceNet
// denoising on MI355X.  Same command line, inputs and outputs as the reference's
// renderer/main_headless.cpp (flags :201-224 + src/opts.cpp:9-29; tree.npz, poses json / txt dir /
// npy, opt.json, ts_*.ts; r_<i>.png or buf_<name>.bin; the 5-line timing report), on top of the
// C ABI in include/rto.h.
//
// Deviations, all documented in INTEGRATION.md:
//   * `--ts_module` is only required when the options say denoise = true (the reference constructs
//     the Denoiser unconditionally and aborts without it, main_headless.cpp:455-456);
//   * TanksAndTemple pose files are read in sorted order (the reference uses directory order);
//   * `--batch B` (1..128, default 100: the reference's 200-pose test loop in two launches) renders B poses per launch of the persistent ray-queue kernel and
//     denoises them as one batch; images are identical to B = 1 (one launch per frame, the reference's loop), the report is still per frame;
//   * extra flags `--shard i/N` (render poses i, i+N, ...: frame sharding across GPUs, one process
//     per GPU), `--gpus N` (this process touches no GPU: it starts N copies of itself, `--shard i/N` on GPU i each,
//     and prints the report of their union) and `--warmup K` (default 100 like the reference).
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "denoiser_torch.h"
#include "imwrite.h"
#include "poses.h"
#include "rto.h"

namespace fs = std::filesystem;
extern char** environ;

namespace {

struct Args {
    std::map<std::string, std::string> kv;
    std::vector<std::string> positional;
    bool has(const std::string& k) const { return kv.count(k) != 0; }
    std::string get(const std::string& k, const std::string& d) const {
        auto it = kv.find(k);
        return it == kv.end() ? d : it->second;
    }
};

// long options taking a value, with their short aliases (opts.cpp:9-29, main_headless.cpp:201-224)
const std::map<std::string, std::string> kShort = {{"w", "width"},      {"h", "height"},     {"s", "step_size"},
                                                   {"e", "stop_thresh"}, {"a", "sigma_thresh"}, {"o", "write_images"},
                                                   {"i", "intrin"},      {"r", "reverse_yz"}};
const char* kFlags[] = {"reverse_yz", "write_buffer", "help", "print_poses", "quant_direct", "torch_net", "fast_filter", "rank_report", "compact_records", "no_denoise_cull"};

bool is_flag(const std::string& k) {
    for (const char* f : kFlags)
        if (k == f) return true;
    return false;
}

Args parse(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string s = argv[i];
        if (s.rfind("--", 0) == 0) {
            std::string k = s.substr(2), v;
            const size_t eq = k.find('=');
            if (eq != std::string::npos) {
                v = k.substr(eq + 1);
                k = k.substr(0, eq);
            } else if (!is_flag(k) && i + 1 < argc) {
                v = argv[++i];
            } else {
                v = "true";
            }
            a.kv[k] = v;
        } else if (s.size() >= 2 && s[0] == '-' && !(s[1] >= '0' && s[1] <= '9') && s[1] != '.') {
            const std::string sk = s.substr(1, 1);
            auto it = kShort.find(sk);
            if (it == kShort.end()) {
                std::fprintf(stderr, "WARNING: unknown option %s ignored\n", s.c_str());  // allow_unrecognised_options
                continue;
            }
            std::string v;
            if (s.size() > 2)
                v = s.substr(2);
            else if (!is_flag(it->second) && i + 1 < argc)
                v = argv[++i];
            else
                v = "true";
            a.kv[it->second] = v;
        } else {
            a.positional.push_back(s);
        }
    }
    return a;
}

void usage() {
    std::puts(
        "Headless PlenOctree volume rendering, MI355X build\n"
        "Usage: volrend_headless npz_file poses [options]\n"
        "  --gpu N            HIP device id (default: current)\n"
        "  -w,--width W  -h,--height H   image size (800x800)\n"
        "  --fx F --fy F      focal lengths (-1 = 1111.11 / fx)\n"
        "  --bg B             background brightness 0-1 (1.0)\n"
        "  -s,--step_size -e,--stop_thresh -a,--sigma_thresh   render options when no --options file\n"
        "  --options opt.json render options (spp, denoise, ...)\n"
        "  --dataset blender|tt|llff (blender)\n"
        "  --ts_module ts.ts  TorchScript GuidanceNet (needed when denoise = true)\n"
        "  -o,--write_images DIR   write r_<i>.png (or buf_<name>.bin with --write_buffer)\n"
        "  --write_buffer  --max_imgs N  --scale S  -i intrin  -r,--reverse_yz\n"
        "  --shard i/N        render poses i, i+N, ... only     --warmup K (100)\n"
        "  --gpus N           frames sharded over N GPUs of this node: N child processes (--shard i/N, GPU i each), one\n"
        "                     report for their union; images do not depend on N (per-pose RNG jump-ahead)\n"
        "  --batch B          poses per launch (1..128, default 100; 1 = one launch per frame like the reference's loop)\n"
        "  --torch_net        run the TorchScript GuidanceNet through libtorch even when it is the compact two-layer\n"
        "                     network the fused HIP kernel implements (default: fused)\n"
        "  --fast_filter      guided filter with factorised exponentials (4 exps per pixel instead of 164; agrees with\n"
        "                     the default, bit-exact form to ~1e-6 relative)\n"
        "  --quant_direct     render a quantised tree.npz from its codebooks (no expansion to dense fp16)\n"
        "  --compact_records  keep SH coefficient records for the leaves of positive density only (half the footprint of a\n"
        "                     dense SH9 / SH16 tree, same pixels, slower shading)\n"
        "  --no_denoise_cull  with --fast_filter and a batch: run GuidanceNet and the filter on every tile, also those that see\n"
        "                     only background (default: such tiles are filled with the kernels' background output; same pixels)\n"
        "  --print_poses      parse the poses, print them (column-major 4x3) and exit\n");
}

// --gpus N: the reference is single-GPU (main_headless.cpp:234-238 picks ONE device); frames are independent, so N
// processes render poses i, i + N, ... each on its own GPU.  This parent never touches a GPU: it starts the
// children (posix_spawn of this very executable: nothing is exec'ed from a process that initialised HIP), relays their
// output and merges their timing reports.  A child that fails makes the whole run fail.
int run_multi_gpu(int argc, char** argv, int gpus) {
    char self[4096];
    const ssize_t sl = readlink("/proc/self/exe", self, sizeof(self) - 1);
    if (sl <= 0) {
        std::fputs("ERROR: cannot resolve /proc/self/exe\n", stderr);
        return 1;
    }
    self[sl] = 0;
    // the devices the children may use: the caller's HIP_VISIBLE_DEVICES list if there is one, else 0..N-1
    std::vector<std::string> visible;
    if (const char* hv = std::getenv("HIP_VISIBLE_DEVICES")) {
        std::string cur;
        for (const char* c = hv;; ++c) {
            if (*c == ',' || *c == 0) {
                if (!cur.empty()) visible.push_back(cur);
                cur.clear();
                if (!*c) break;
            } else {
                cur.push_back(*c);
            }
        }
    }
    struct Child {
        pid_t pid = -1;
        int fd = -1;
        std::string out;
        int status = -1;
    };
    std::vector<Child> kids((size_t)gpus);
    for (int r = 0; r < gpus; ++r) {
        std::vector<std::string> av;
        av.push_back(self);
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "--gpus" || a == "--gpu") {  // replaced below
                ++i;
                continue;
            }
            if (a.rfind("--gpus=", 0) == 0 || a.rfind("--gpu=", 0) == 0) continue;
            av.push_back(a);
        }
        av.push_back("--shard");
        av.push_back(std::to_string(r) + "/" + std::to_string(gpus));
        av.push_back("--gpu");
        av.push_back("0");  // (the one device its HIP_VISIBLE_DEVICES shows)
        av.push_back("--rank_report");
        std::vector<char*> cav;
        for (auto& a : av) cav.push_back(const_cast<char*>(a.c_str()));
        cav.push_back(nullptr);
        std::vector<std::string> env;
        for (char** e = environ; *e; ++e)
            if (std::strncmp(*e, "HIP_VISIBLE_DEVICES=", 20) != 0) env.push_back(*e);
        env.push_back("HIP_VISIBLE_DEVICES=" + ((size_t)r < visible.size() ? visible[(size_t)r] : std::to_string(r)));
        std::vector<char*> cenv;
        for (auto& e : env) cenv.push_back(const_cast<char*>(e.c_str()));
        cenv.push_back(nullptr);
        int pfd[2];
        if (pipe(pfd) != 0) {
            std::perror("pipe");
            return 1;
        }
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        posix_spawn_file_actions_adddup2(&fa, pfd[1], STDOUT_FILENO);
        posix_spawn_file_actions_addclose(&fa, pfd[0]);
        posix_spawn_file_actions_addclose(&fa, pfd[1]);
        const int rc = posix_spawn(&kids[(size_t)r].pid, self, &fa, nullptr, cav.data(), cenv.data());
        posix_spawn_file_actions_destroy(&fa);
        close(pfd[1]);
        if (rc != 0) {
            std::fprintf(stderr, "ERROR: cannot start rank %d: %s\n", r, std::strerror(rc));
            return 1;
        }
        kids[(size_t)r].fd = pfd[0];
    }
    std::vector<std::thread> readers;
    for (auto& k : kids)
        readers.emplace_back([&k]() {
            char buf[4096];
            ssize_t n;
            while ((n = read(k.fd, buf, sizeof(buf))) > 0) k.out.append(buf, (size_t)n);
            close(k.fd);
            waitpid(k.pid, &k.status, 0);
        });
    for (auto& t : readers) t.join();
    bool ok = true;
    double sum_ms[3] = {0, 0, 0}, sum_fps = 0;
    long frames = 0;
    for (int r = 0; r < gpus; ++r) {
        const Child& k = kids[(size_t)r];
        if (!WIFEXITED(k.status) || WEXITSTATUS(k.status) != 0) {
            std::fprintf(stderr, "ERROR: rank %d (GPU %d) failed (status %d)\n", r, r, k.status);
            ok = false;
        }
        size_t pos = 0;
        while (pos < k.out.size()) {  // relay everything but the machine-readable line, tagged with the rank
            size_t e = k.out.find('\n', pos);
            if (e == std::string::npos) e = k.out.size();
            const std::string line = k.out.substr(pos, e - pos);
            pos = e + 1;
            double a, b, c;
            long n;
            if (std::sscanf(line.c_str(), "RANK_REPORT %lf %lf %lf %ld", &a, &b, &c, &n) == 4) {
                sum_ms[0] += a * n;
                sum_ms[1] += b * n;
                sum_ms[2] += c * n;
                frames += n;
                if (a + b + c > 0) sum_fps += 1000.0 / (a + b + c);
            } else if (!line.empty()) {
                std::printf("[rank %d] %s\n", r, line.c_str());
            }
        }
    }
    if (!ok) return 1;
    if (frames > 0) {  // Timer::report (render_context.hpp:190-206) over the union of the ranks' frames
        const double m0 = sum_ms[0] / frames, m1 = sum_ms[1] / frames, m2 = sum_ms[2] / frames;
        std::printf("render: %.10f ms per frame\n", m0);
        std::printf("torch:  %.10f ms per frame\n", m1);
        std::printf("filter: %.10f ms per frame\n", m2);
        std::printf("all:    %.10f ms per frame\n", m0 + m1 + m2);
        std::printf("FPS:    %.10f\n", sum_fps);
        std::printf("INFO: %ld frames on %d GPUs; FPS = the sum of the ranks' 1000 / (render + torch + filter)\n", frames, gpus);
    }
    return 0;
}

#define CHECK_RTO(expr)                                                          \
    do {                                                                         \
        if ((expr) != RTO_OK) {                                                  \
            std::fprintf(stderr, "ERROR: %s: %s\n", #expr, rto_last_error());    \
            return 1;                                                            \
        }                                                                        \
    } while (0)

}  // namespace

int main(int argc, char** argv) {
    Args args = parse(argc, argv);
    // `--file tree.npz` names the tree like the first positional argument does (opts.cpp:12,36: parse_positional({"file"}));
    // `--draw` (a GUI draw list) is accepted and ignored
    if (args.has("file")) args.positional.insert(args.positional.begin(), args.get("file", ""));
    if (args.has("help") || args.positional.size() < 2) {
        usage();
        return args.has("help") ? 0 : 1;
    }
    {
        const int gpus = std::atoi(args.get("gpus", "1").c_str());
        if (gpus < 1 || gpus > 64) {
            std::fputs("ERROR: --gpus expects 1..64\n", stderr);
            return 1;
        }
        if (gpus > 1) {
            if (args.has("shard")) {
                std::fputs("ERROR: --gpus N starts the shards itself; give one of --gpus and --shard\n", stderr);
                return 1;
            }
            return run_multi_gpu(argc, argv, gpus);  // before anything that could touch a GPU
        }
    }
    const std::string tree_path = args.positional[0], poses_path = args.positional[1];
    const int device = std::max(0, std::atoi(args.get("gpu", "-1").c_str()));

    rto::PoseSet ps;
    ps.width = std::atoi(args.get("width", "800").c_str());
    ps.height = std::atoi(args.get("height", "800").c_str());
    ps.fx = (float)std::atof(args.get("fx", "-1.0").c_str());
    if (ps.fx < 0) ps.fx = 1111.11f;  // main_headless.cpp:241-243
    ps.fy = (float)std::atof(args.get("fy", "-1.0").c_str());
    if (ps.fy < 0) ps.fy = ps.fx;
    const std::string dataset = args.get("dataset", "blender");
    try {
        rto::load_poses(dataset, poses_path, args.has("reverse_yz"), ps);
        if (args.has("intrin") && !args.get("intrin", "").empty()) {  // -i overrides fx/fy
            std::ifstream f(args.get("intrin", ""));
            float _;
            if (f) {
                f >> ps.fx >> _ >> _ >> _;
                f >> _ >> ps.fy;
            }
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    if (ps.trans.empty()) {
        std::fputs("WARNING: No camera poses specified, quitting\n", stderr);
        return 1;
    }
    int shard_i = 0, shard_n = 1;
    if (args.has("shard")) {
        if (std::sscanf(args.get("shard", "0/1").c_str(), "%d/%d", &shard_i, &shard_n) != 2 || shard_n < 1 ||
            shard_i < 0 || shard_i >= shard_n) {
            std::fputs("ERROR: --shard expects i/N with 0 <= i < N\n", stderr);
            return 1;
        }
    }
    if (args.has("print_poses")) {  // host-only check of the pose loaders (and of the sharding): no tree, no device
        std::printf("POSES %zu %d %d %.9g %.9g\n", ps.trans.size(), ps.width, ps.height, ps.fx, ps.fy);
        for (size_t i = 0; i < ps.trans.size(); ++i) {
            if ((int)(i % shard_n) != shard_i) continue;
            std::printf("%s", ps.basenames[i].c_str());
            for (float v : ps.trans[i]) std::printf(" %.9g", v);
            std::printf("\n");
        }
        return 0;
    }

    rto_tree* tree = nullptr;
    CHECK_RTO(rto_tree_load_npz_ex(tree_path.c_str(), device,
                                   (args.has("quant_direct") ? RTO_TREE_QUANT_DIRECT : 0) |
                                       (args.has("compact_records") ? RTO_TREE_COMPACT_RECORDS : 0),
                                   &tree));
    if (dataset == "llff") CHECK_RTO(rto_tree_set_ndc(tree, (float)ps.width, (float)ps.height, ps.fx));  // :400-405

    int width = ps.width, height = ps.height;
    float fx = ps.fx, fy = ps.fy;
    {
        const float scale = (float)std::atof(args.get("scale", "1").c_str());  // :407-417
        if (scale != 1.f) {
            const int ow = width, oh = height;
            width = (int)(width * scale);
            height = (int)(height * scale);
            fx *= (float)width / ow;
            fy *= (float)height / oh;
        }
    }
    {
        const int max_imgs = std::atoi(args.get("max_imgs", "0").c_str());  // :419-426
        if (max_imgs > 0 && ps.trans.size() > (size_t)max_imgs) {
            ps.trans.resize(max_imgs);
            ps.basenames.resize(max_imgs);
        }
    }
    const std::string out_dir = args.get("write_images", "");
    if (!out_dir.empty()) fs::create_directories(out_dir);

    rto_options options;
    rto_options_default(&options);
    const std::string options_path = args.get("options", "");
    if (!options_path.empty()) {
        CHECK_RTO(rto_options_from_json_file(options_path.c_str(), &options));
    } else {  // render_options_from_args opts.cpp:44-66
        options.background_brightness = (float)std::atof(args.get("bg", "1.0").c_str());
        options.step_size = (float)std::atof(args.get("step_size", "1e-4").c_str());
        options.stop_thresh = (float)std::atof(args.get("stop_thresh", "1e-2").c_str());
        options.sigma_thresh = (float)std::atof(args.get("sigma_thresh", "1e-2").c_str());
    }

    int batch = std::max(1, std::min(128, std::atoi(args.get("batch", "100").c_str())));
    const int filter_mode = args.has("fast_filter") ? RTO_FILTER_FACTORISED : RTO_FILTER_EXACT;
    {  // no more frame slots than this process has poses to render
        const size_t n_mine = (ps.trans.size() + (size_t)shard_n - 1 - (size_t)shard_i) / (size_t)shard_n;
        if ((size_t)batch > n_mine) batch = (int)std::max<size_t>(1, n_mine);
        // the batched kernels address their hit lists with 32-bit offsets: frames x spp x pixels < 2^32
        const uint64_t per_frame = (uint64_t)std::max(1, options.spp) * (uint64_t)width * (uint64_t)height;
        const uint64_t cap = 0xffffffffULL / std::max<uint64_t>(1, per_frame);
        if ((uint64_t)batch > cap) batch = (int)std::max<uint64_t>(1, cap);
    }
    rto_ctx* ctx = nullptr;
    CHECK_RTO(rto_ctx_create_batch(width, height, batch, device, &ctx));

    std::unique_ptr<rto::TorchDenoiser> denoiser;
    if (options.denoise) {
        try {
            denoiser = std::make_unique<rto::TorchDenoiser>(args.get("ts_module", ""), device, !args.has("torch_net"));
            std::printf("INFO: GuidanceNet runs %s\n", denoiser->fused() ? "as the fused HIP kernel" : "through libtorch");
        } catch (const std::exception& e) {
            std::fprintf(stderr, "ERROR: %s\n", e.what());
            return 1;
        }
    }

    rto_camera cam;
    cam.width = width;
    cam.height = height;
    cam.fx = fx;
    cam.fy = fy;
    void* stream = nullptr;  // the default stream: libtorch's current stream in this process

    // Denoiser::denoise (denoiser.cpp:31-61) for the n frames in context slots 0..n-1 (the selected slot must be 0 for
    // n > 1).  Fused network + --fast_filter: the maps stay packed fp16 between the two kernels (same pixels as fp32 maps).
    // after_batch: the frames are those of the rto_launch_renderer_batch just issued -- its tile marks tell both kernels which
    // of their tiles see nothing but background (filled, not computed: same bits; --no_denoise_cull computes them all)
    const bool denoise_cull = !args.has("no_denoise_cull");
    auto denoise_n = [&](int n, bool timed, bool after_batch = false) -> int {
        const float *w = nullptr, *g = nullptr;
        int L = 0;
        int rc = RTO_OK;
        const uint32_t* marks = nullptr;
        int mark_words = 0, mark_slot = 0, mark_frames = 0;
        float mark_bg = 0.f;
        if (after_batch && denoise_cull &&
            (rto_ctx_tile_marks(ctx, &marks, &mark_words, &mark_slot, &mark_frames, &mark_bg) != RTO_OK || mark_slot != 0 || mark_frames < n))
            marks = nullptr;
        if (timed) rto_timer_start(ctx, RTO_T_TORCH);
        const bool packed = denoiser->fused() && filter_mode == RTO_FILTER_FACTORISED;
        // frames of a lean batched launch (below): no aux planes, the noisy image holds (r, g, b, alpha) -- the network reads that
        const int lean_state = rto_ctx_frames_are_lean(ctx, 0, n);
        if (lean_state < 0) return RTO_E_INVALID;  // (cannot happen here: every batch rewrites all the slots it denoises)
        const bool lean = lean_state == 1;
        if (packed)
            rc = rto_guidance_net_forward_packed_culled(denoiser->fused_handle(), stream, lean ? rto_ctx_noisy(ctx) : rto_ctx_aux(ctx), n, height,
                                                        width, lean ? RTO_NET_INPUT_RGBA : RTO_NET_AUX_SQUARES_IMPLIED, marks, mark_words, mark_bg);
        else
            denoiser->forward(lean ? rto_ctx_noisy(ctx) : rto_ctx_aux(ctx), n, height, width, &w, &g, &L, lean);
        if (timed) rto_timer_stop(ctx, RTO_T_TORCH);
        if (rc != RTO_OK) return rc;
        if (timed) rto_timer_start(ctx, RTO_T_FILTER);
        rc = packed ? rto_filtering_packed_culled(denoiser->fused_handle(), stream, rto_ctx_noisy(ctx), rto_ctx_image(ctx), n, height, width,
                                                  marks, mark_words, mark_bg)
                    : rto_filtering_batch_mode(stream, w, g, L, height, width, n, rto_ctx_noisy(ctx), rto_ctx_image(ctx), filter_mode);
        if (timed) rto_timer_stop(ctx, RTO_T_FILTER);
        return rc;
    };
    auto denoise = [&]() -> int { return denoise_n(1, true); };

    // Warm up (main_headless.cpp:469-479): pose 0, every iteration advances the RNG
    rto_timer_reset(ctx, stream);
    const int warmup = std::atoi(args.get("warmup", "100").c_str());
    std::memcpy(cam.transform, ps.trans[0].data(), sizeof(cam.transform));
    for (int i = 0; i < warmup; ++i) {
        CHECK_RTO(rto_launch_renderer(tree, &cam, &options, ctx, stream));
        if (options.denoise) CHECK_RTO(denoise());
        rto_ctx_rng_advance(ctx, 1LL << 32);
    }
    if (batch > 1 && options.denoise && warmup > 0) {
        // the batched network forward has its own shapes (convolution algorithm search, allocator
        // growth in libtorch): warm those up as well, like the single-frame denoise above
        const size_t n_mine = (ps.trans.size() + (size_t)shard_n - 1 - (size_t)shard_i) / (size_t)shard_n;
        const int shapes[2] = {(int)std::min<size_t>((size_t)batch, n_mine), (int)(n_mine % (size_t)batch)};
        for (int n : shapes) {
            if (n < 1) continue;
            for (int rep = 0; rep < 3; ++rep) CHECK_RTO(denoise_n(n, false));
        }
    }
    rto_timer_reset(ctx, stream);

    std::vector<uint8_t> rgba8((size_t)width * height * 4);
    std::vector<float> aux;
    const bool write_buffer = args.has("write_buffer");
    if (write_buffer) aux.resize((size_t)width * height * RTO_AUX_CHANNELS);
    // Batched launches whose aux planes nobody reads store the 16 bytes per pixel the fused denoise stage consumes instead of the
    // reference's 48 (rto_ctx_set_lean_outputs; same PNGs).  --write_buffer dumps the aux planes, the TorchScript module reads them.
    if (batch > 1 && options.denoise && denoiser && denoiser->fused() && !write_buffer) CHECK_RTO(rto_ctx_set_lean_outputs(ctx, 1));

    if (batch > 1) {
        // throughput form of the loop below: groups of `batch` poses per launch.  ctx.rng stays at its
        // post-warm-up state; pose i uses it advanced i times (explicit jump counts), exactly the
        // state the per-frame loop reaches.
        std::vector<size_t> mine;
        for (size_t i = 0; i < ps.trans.size(); ++i)
            if ((int)(i % shard_n) == shard_i) mine.push_back(i);
        std::vector<rto_camera> cams(batch, cam);
        std::vector<int64_t> jumps(batch);
        size_t rendered = 0;
        int groups = 0;
        for (size_t g0 = 0; g0 < mine.size(); g0 += batch, ++groups) {
            const int n = (int)std::min<size_t>(batch, mine.size() - g0);
            for (int f = 0; f < n; ++f) {
                std::memcpy(cams[f].transform, ps.trans[mine[g0 + f]].data(), sizeof(cam.transform));
                jumps[f] = (int64_t)mine[g0 + f];
            }
            rto_ctx_select_frame(ctx, 0);
            rto_timer_start(ctx, RTO_T_RENDER);
            CHECK_RTO(rto_launch_renderer_batch(tree, cams.data(), jumps.data(), n, &options, ctx, stream));
            rto_timer_stop(ctx, RTO_T_RENDER);
            if (options.denoise) CHECK_RTO(denoise_n(n, true, true));
            CHECK_RTO(rto_timer_record(ctx, options.denoise));
            rendered += (size_t)n;
            if (out_dir.empty()) continue;
            for (int f = 0; f < n; ++f) {
                const size_t i = mine[g0 + f];
                rto_ctx_select_frame(ctx, f);
                if (write_buffer) {
                    CHECK_RTO(rto_ctx_download_aux(ctx, stream, aux.data()));
                    std::ofstream fo(out_dir + "/buf_" + ps.basenames[i] + ".bin", std::ios::binary);
                    fo.write(reinterpret_cast<const char*>(aux.data()), (std::streamsize)(aux.size() * sizeof(float)));
                } else {
                    CHECK_RTO(rto_ctx_download_rgba8(ctx, stream, 0, rgba8.data()));
                    if (!rto::write_png_rgba8(out_dir + "/" + ps.basenames[i] + ".png", rgba8.data(), width, height)) {
                        std::fprintf(stderr, "ERROR: cannot write %s/%s.png\n", out_dir.c_str(), ps.basenames[i].c_str());
                        return 1;
                    }
                }
            }
        }
        float ms[3], fps;
        int cnt;
        rto_timer_report(ctx, ms, &fps, &cnt);
        const float per_frame = rendered ? (float)groups / (float)rendered : 0.f;  // group means -> per frame
        for (float& m : ms) m *= per_frame;
        const float all = ms[0] + ms[1] + ms[2];
        std::printf("render: %.10f ms per frame\n", ms[0]);
        std::printf("torch:  %.10f ms per frame\n", ms[1]);
        std::printf("filter: %.10f ms per frame\n", ms[2]);
        std::printf("all:    %.10f ms per frame\n", all);
        std::printf("FPS:    %.10f\n", all > 0 ? 1000.f / all : 0.f);
        if (args.has("rank_report")) std::printf("RANK_REPORT %.10f %.10f %.10f %zu\n", ms[0], ms[1], ms[2], rendered);
        denoiser.reset();
        rto_ctx_free(ctx);
        rto_tree_free(tree);
        return 0;
    }

    for (size_t i = 0; i < ps.trans.size(); ++i) {  // :485-543
        if ((int)(i % shard_n) == shard_i) {
            std::memcpy(cam.transform, ps.trans[i].data(), sizeof(cam.transform));
            rto_timer_start(ctx, RTO_T_RENDER);
            CHECK_RTO(rto_launch_renderer(tree, &cam, &options, ctx, stream));
            rto_timer_stop(ctx, RTO_T_RENDER);
            if (options.denoise) CHECK_RTO(denoise());
            CHECK_RTO(rto_timer_record(ctx, options.denoise));
        }
        rto_ctx_rng_advance(ctx, 1LL << 32);  // skipped poses advance too: images do not depend on N
        if ((int)(i % shard_n) != shard_i || out_dir.empty()) continue;
        if (write_buffer) {
            CHECK_RTO(rto_ctx_download_aux(ctx, stream, aux.data()));
            std::ofstream f(out_dir + "/buf_" + ps.basenames[i] + ".bin", std::ios::binary);
            f.write(reinterpret_cast<const char*>(aux.data()), (std::streamsize)(aux.size() * sizeof(float)));
        } else {
            CHECK_RTO(rto_ctx_download_rgba8(ctx, stream, 0, rgba8.data()));
            if (!rto::write_png_rgba8(out_dir + "/" + ps.basenames[i] + ".png", rgba8.data(), width, height)) {
                std::fprintf(stderr, "ERROR: cannot write %s/%s.png\n", out_dir.c_str(), ps.basenames[i].c_str());
                return 1;
            }
        }
    }

    float ms[3], fps;  // Timer::report render_context.hpp:190-206
    int frames;
    rto_timer_report(ctx, ms, &fps, &frames);
    std::printf("render: %.10f ms per frame\n", ms[0]);
    std::printf("torch:  %.10f ms per frame\n", ms[1]);
    std::printf("filter: %.10f ms per frame\n", ms[2]);
    std::printf("all:    %.10f ms per frame\n", ms[0] + ms[1] + ms[2]);
    std::printf("FPS:    %.10f\n", fps);
    if (args.has("rank_report")) std::printf("RANK_REPORT %.10f %.10f %.10f %d\n", ms[0], ms[1], ms[2], frames);

    denoiser.reset();
    rto_ctx_free(ctx);
    rto_tree_free(tree);
    return 0;
}
