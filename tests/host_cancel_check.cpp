// AsyncSubFilter::cancel (demiurge_amd/host/Filter.h) without a GPU: a worker that would run for ever ends at its next loop
// check; a worker parked in dispatchGPU is released with its closure dropped; the destructor cancels before it joins.
#include <chrono>
#include <cstdio>
#include "Filter.h"

struct Endless : AsyncSubFilter {
    std::atomic<int> rounds{0}, closures{0};
    bool park;                                   // hand a closure to the "GL" thread every round
    explicit Endless(bool p) : park(p) {}
    void run() override
    {
        for (int j = 0; j < 1000000000 && !cancelled(); j++) {
            if (park) dispatchGPU([this](Project *) { closures++; });
            else std::this_thread::sleep_for(std::chrono::milliseconds(1));
            rounds++;
            setProgress({ false, 0.5f });
        }
        setProgress({ true, 1.0f });
    }
};

int main()
{
    Project p;
    auto t0 = std::chrono::steady_clock::now();
    {   // cancelled while being stepped: finishes within a few frames
        Endless f(true);
        int frames = 0;
        for (; frames < 100000; frames++) {
            if (frames == 20) f.cancel();
            if (f.step(&p).first) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (frames >= 100000 || f.closures.load() < 5) { printf("FAIL stepped: frames %d closures %d\n", frames, f.closures.load()); return 1; }
        printf("stepped: finished %d frames after cancel, %d closures run\n", frames - 20, f.closures.load());
    }
    {   // destroyed while the worker is parked in dispatchGPU and nobody steps any more
        auto *f = new Endless(true);
        for (int i = 0; i < 5; i++) { f->step(&p); std::this_thread::sleep_for(std::chrono::milliseconds(2)); }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));      // the worker has posted its next closure by now
        delete f;
        printf("destroyed while parked\n");
    }
    {   // destroyed in the middle of a compute loop
        auto *f = new Endless(false);
        f->step(&p);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        delete f;
        printf("destroyed while computing\n");
    }
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (s > 5.0) { printf("FAIL: %.1f s\n", s); return 1; }
    printf("ok %.2f s\n", s);
    return 0;
}
