// AsyncSubFilter::cancel (demiurge_amd/host/Filter.h) without a GPU: a worker that would run for ever ends at its next loop
// check; a worker parked in dispatchGPU is released with its closure dropped; the destructor cancels before it joins.
#include <chrono>
#include <cstdio>
#include <memory>
#include "Filter.h"

struct Endless : AsyncSubFilter {
    std::atomic<int> rounds{0}, closures{0};
    bool park;                                   // hand a closure to the "GL" thread every round
    explicit Endless(bool p) : park(p) {}
    ~Endless() override { stop(); }              // (the join comes before the members -- and the vptr -- go: Filter.h)
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
    {   // ProgressFilter / BackupFilter (reference Filter.cpp:12-23,70-115): a tool that rewrites the terrain every round, cancelled in
        // mid-run -- the terrain is the backup again, finalizeFilter has been called once, and no parked upload lands afterwards
        struct Painter : AsyncSubFilter {
            ~Painter() override { stop(); }
            std::vector<float> mine;                       // (a member the worker reads: stop() must join before it goes)
            void run() override
            {
                for (int j = 1; j < 1000000 && !cancelled(); j++) {
                    mine.assign(64, (float)j);
                    dispatchGPU([this](Project *q) { q->terrain = mine; });
                    setProgress({ false, j / 1000.0f });
                }
                setProgress({ true, 1.0f });
            }
        };
        Project q;
        q.width = 8; q.height = 8; q.terrain.assign(64, -1.0f);
        auto pf = std::make_unique<ProgressFilter>(&q, [](Project *x) { return &x->terrain; }, std::make_unique<Painter>());
        int frames = 0;
        while (!pf->isFinished() && frames < 100000) {
            pf->run(&q);
            if (++frames == 30) pf->abort();
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        const bool painted = pf->getProgress() > 0.0f;
        bool restored = pf->isFinished() && pf->wasAborted() && q.finalized == 1;
        for (float v : q.terrain) restored = restored && v == -1.0f;
        pf->run(&q);                                       // a finished filter does nothing more
        pf.reset();                                        // joins the worker
        for (float v : q.terrain) restored = restored && v == -1.0f;
        if (!painted || !restored) { printf("FAIL progress filter: painted %d restored %d finalized %d\n", (int)painted, (int)restored, q.finalized); return 1; }
        // restoreUnselected: the result where the selection is 1, the backup where it is 0, a blend in between (Filter.cpp:52-68)
        Project r;
        r.width = 2; r.height = 1; r.terrain = { 1.0f, 1.0f }; r.selection = { 1.0f, 0.25f };
        struct Plain : BackupFilter { using BackupFilter::BackupFilter; void run(Project *) override {} bool isFinished() override { return true; } };
        Plain bf(&r, [](Project *x) { return &x->terrain; });
        r.terrain = { 5.0f, 5.0f };
        bf.restoreUnselected();
        if (r.terrain[0] != 5.0f || r.terrain[1] != 0.25f * 5.0f + 0.75f * 1.0f) { printf("FAIL restoreUnselected: %g %g\n", r.terrain[0], r.terrain[1]); return 1; }
        printf("progress filter: cancelled after %d frames, terrain restored, finalized once\n", frames);
    }
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (s > 5.0) { printf("FAIL: %.1f s\n", s); return 1; }
    printf("ok %.2f s\n", s);
    return 0;
}
