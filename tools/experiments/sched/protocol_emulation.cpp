// The segment scheduler's take / give protocol (segment_scheduler_v3 / v4 patches: sched_take, sched_give, the "unfinished" counters)
// restated with std::atomic and run by host threads under a model of the GPU's dispatcher — 8 XCDs x SLOTS resident workgroups,
// workgroups issued IN ORDER, workgroup b on XCD b mod 8, the dispatcher waiting whenever that XCD has no free slot.  It checks the
// protocol's logic, not the hardware: does every chain make all its transitions exactly once, does the launch end, how many
// workgroups does it use, with randomised segment times.
//     g++ -O2 -std=c++17 -pthread -o /tmp/protocol_emulation protocol_emulation.cpp && /tmp/protocol_emulation [v3|v4] [chains] [N] [quantum]
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

static const int XCDS = 8;
static int SLOTS = 16;                       // per XCD (128 on the device; fewer here: host threads)
static int C, N, Q;
static bool V4;
static std::vector<std::atomic<unsigned>> ctr(320);
static std::vector<std::atomic<int>> ring, done_arr, made;     // made[c]: transitions actually run (checked at the end)
static std::vector<int> order;
static const int LONG = 1 << 30;
static std::atomic<long> polls{0};

static void nap_us(int us) { std::this_thread::sleep_for(std::chrono::microseconds(us)); }

static int sched_take(unsigned x) {
    std::atomic<unsigned>&head = ctr[32 + 32 * x], &tail = ctr[48 + 32 * x];
    const int limit = V4 ? 64 : (1 << 22);
    for (int poll = 0; poll < limit; ++poll) {
        if (!V4 && ctr[1].load() == 0u) return -1;
        if (ctr[0].load() < (unsigned)C) {
            const unsigned f = ctr[0].fetch_add(1u);
            if (f < (unsigned)C) { if (V4) ctr[2 + x].fetch_add(1u); return order[f]; }
        }
        unsigned h = head.load(), t = tail.load();
        if (h < t) {
            if (!head.compare_exchange_strong(h, h + 1)) continue;
            std::atomic<int>& slot = ring[(size_t)x * C + h % (unsigned)C];
            int c;
            while ((c = slot.load()) < 0) std::this_thread::yield();
            slot.store(-1);
            return c;
        }
        if (V4 && ctr[2 + x].load() == 0u) return -1;
        polls++;
        nap_us(V4 ? 1 : 20);
    }
    return -1;
}

static void sched_give(unsigned x, int chain, int done) {
    done_arr[chain].store(done);
    const unsigned t = ctr[48 + 32 * x].fetch_add(1u);
    ring[(size_t)x * C + t % (unsigned)C].store(chain);
}

struct Xcd { std::mutex m; std::condition_variable cv; int free_slots; };
static Xcd xcd[XCDS];

static void workgroup(unsigned x, unsigned seed) {
    const int item = sched_take(x);
    if (item >= 0) {
        const int chain = item & ~LONG;
        const int n_off = done_arr[chain].load();
        const int count = ((item & LONG) || N - n_off <= Q + Q / 2) ? N - n_off : Q;
        std::minstd_rand rng(seed);
        nap_us(20 + (int)(count * (1 + rng() % 3)));            // the segment: ≈ 1–3 µs per transition here
        made[chain].fetch_add(count);
        if (n_off + count < N) sched_give(x, chain, n_off + count);
        else { done_arr[chain].store(N); ctr[1].fetch_sub(1u); if (V4) ctr[2 + x].fetch_sub(1u); }
    }
    { std::lock_guard<std::mutex> g(xcd[x].m); xcd[x].free_slots++; }
    xcd[x].cv.notify_one();
}

int main(int argc, char** argv) {
    V4 = argc > 1 && !strcmp(argv[1], "v4");
    C = argc > 2 ? atoi(argv[2]) : 512; N = argc > 3 ? atoi(argv[3]) : 200; Q = argc > 4 ? atoi(argv[4]) : 16;
    if (argc > 5) SLOTS = atoi(argv[5]);
    ring = std::vector<std::atomic<int>>((size_t)XCDS * C); done_arr = std::vector<std::atomic<int>>(C); made = std::vector<std::atomic<int>>(C);
    for (auto& r : ring) r.store(-1);
    for (int c = 0; c < C; ++c) { done_arr[c].store(0); made[c].store(0); order.push_back(c < 3 ? (c | LONG) : c); }
    for (auto& a : ctr) a.store(0);
    ctr[1].store(C);
    int nseg = 0;
    for (int left = N; left > 0; ++nseg) left -= (left <= Q + Q / 2) ? left : Q;
    const long total = 3 + (long)(C - 3) * nseg, grid = V4 ? 8 * total + 8192 : 2 * total + 2048;
    for (int x = 0; x < XCDS; ++x) xcd[x].free_slots = SLOTS;
    const auto t0 = std::chrono::steady_clock::now();
    long issued = 0, launches = 0;
    std::vector<std::thread> threads;
    while (ctr[1].load() != 0u && launches < 9) {
        ++launches;
        for (long b = 0; b < grid; ++b) {
            const unsigned x = b % XCDS;
            {   // in-order issue: wait for a slot on THIS workgroup's XCD
                std::unique_lock<std::mutex> g(xcd[x].m);
                if (!xcd[x].cv.wait_for(g, std::chrono::seconds(20), [&] { return xcd[x].free_slots > 0; })) { printf("dispatcher blocked for 20 s on XCD %u: STALL\n", x); return 1; }
                xcd[x].free_slots--;
            }
            threads.emplace_back(workgroup, x, (unsigned)(b * 2654435761u));
            ++issued;
            if (threads.size() > 4096) { for (auto& t : threads) t.join(); threads.clear(); }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { printf("time limit: STALL (unfinished %u)\n", ctr[1].load()); return 1; }
        }
        for (auto& t : threads) t.join();
        threads.clear();
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    long bad = 0;
    for (int c = 0; c < C; ++c) bad += made[c].load() != N;
    printf("%s: %d chains x %d transitions, quantum %d, %d slots per XCD: %ld workgroups issued of %ld per launch, %ld launch(es), %.0f ms, polls %ld, unfinished %u, chains with a wrong transition count %ld\n",
           V4 ? "v4" : "v3", C, N, Q, SLOTS, issued, grid, launches, ms, polls.load(), ctr[1].load(), bad);
    return bad || ctr[1].load() ? 1 : 0;
}
