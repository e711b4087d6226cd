// tools/ubench/decide_trace.cpp -- where the time of ONE tdt_decide launch goes (exact form, one utterance, vocabulary 1024 + 5 durations, frame window 8):
// phase stamps of thread 0 (decode_dev.hpp, -DDEC_TRACE) for (a) a token decision, (b) a blank walked into a token, (c) blanks until the window runs out,
// and the launch's duration by hipEvents over 200 launches.     make -C tools/ubench decide_trace ; tools/ubench/decide_trace
#define DEC_TRACE 1
#include "../../parakeet.cpp_amd/csrc/kernels/decode.hip"
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using namespace pk;

template <typename T> static T *dalloc(size_t n) { T *p = nullptr; if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) { printf("alloc failed\n"); exit(2); } (void)hipMemset(p, 0, n * sizeof(T)); return p; }

int main(int argc, char **argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 8;
    const int V = 1025, D = 5, VD = V + D, Hp = 640, J = 640, T = 4096, max_tokens = 4096;
    TdtState st{};
    st.B = 1; st.T = T; st.V = V; st.D = D; st.L = 1; st.Hp = Hp; st.blank = 1024; st.max_symbols = 10; st.max_tokens = max_tokens; st.max_steps = 1 << 20;
    for (int i = 0; i < 5; ++i) st.durations[i] = i;
    float *logits = dalloc<float>((size_t)8 * VD);
    st.logits = logits;
    st.h = dalloc<float>(Hp); st.c = dalloc<float>(Hp); st.hn = dalloc<float>(Hp); st.cn = dalloc<float>(Hp);
    int *ib = dalloc<int>(64);
    st.token = ib; st.t = ib + 1; st.nsym = ib + 2; st.n_out = ib + 3; st.steps = ib + 4; st.done = ib + 5; st.done_count = ib + 6; st.need = ib + 16;
    st.lens = dalloc<int>(1); st.ids = dalloc<int>(max_tokens); st.start = dalloc<int>(max_tokens); st.end = dalloc<int>(max_tokens); st.conf = dalloc<float>(max_tokens);
    st.margin = dalloc<float>(1);
    st.pp = dalloc<float>(J); st.ep = dalloc<float>((size_t)T * J); st.z = dalloc<float>((size_t)8 * J); st.J = J;
    st.F = F;
    long long *dtr = dalloc<long long>(16), *null = nullptr, h[16];
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<float> hl((size_t)8 * VD);
    srand(3);
    const char *names[3] = {"token at once", "one blank (duration 1), then a token", "blanks of duration 1 until the window runs out"};
    for (int scen = 0; scen < 3; ++scen) {
        for (auto &v : hl) v = (float)rand() / RAND_MAX;
        for (int f = 0; f < 8; ++f) {
            const bool blank = scen == 2 || (scen == 1 && f == 0);
            hl[(size_t)f * VD + (blank ? 1024 : 17)] = 9.0f;       // label head
            hl[(size_t)f * VD + V + 1] = 9.0f;                     // duration 1
        }
        CK(hipMemcpy(logits, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
        double ev_us = 0.0;
        for (int pass = 0; pass < 2; ++pass) {                     // pass 0: stamps of one launch; pass 1: 200 launches by events (state reset by tdt_init each time would cost a launch: t just runs on)
            launch_tdt_init(st, s);
            CK(hipStreamSynchronize(s));
            if (pass == 0) {
                CK(hipMemset(dtr, 0, sizeof h));
                CK(hipMemcpyToSymbol(HIP_SYMBOL(dec_trace), &dtr, 8));
                launch_tdt_decide(st, s);
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(h, dtr, sizeof h, hipMemcpyDeviceToHost));
                CK(hipMemcpyToSymbol(HIP_SYMBOL(dec_trace), &null, 8));
            } else {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int i = 0; i < 20; ++i) launch_tdt_decide(st, s);
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 200; ++i) launch_tdt_decide(st, s);
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                ev_us = ms * 1e3 / 200;
            }
        }
        printf("F = %d, %s: %lld decision(s) in the launch; %.2f us per launch (events, back to back)\n", F, names[scen], h[15], ev_us);
        const char *lab[11] = {"entry", "LAST decision starts", "row staged", "maximum known", "exps in LDS", "wave 0: sum + log", "lse known (barrier)", "argmax exchanged",
                               "walk over", "commit / next z issued", "state words issued"};
        printf("   first decision starts at +%lld clocks\n", h[11] - h[0]);
        for (int i = 1; i <= 10; ++i) printf("   %-26s +%6lld clocks (%+6lld)\n", lab[i], h[i] - h[0], h[i] - h[i - 1]);
    }
    return 0;
}
