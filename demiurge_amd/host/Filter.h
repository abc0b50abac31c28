// GL-free mirror of the reference's operator surface (src/filter/Filter.h:58-115, src/filter/Filter.cpp:125-166),
// so that adapters written for the editor can be built and tested here without imgui/glfw/GL.
//
//   SubFilter::step(Project*) -> {finished, progress}       the editor calls it once per UI frame
//   AsyncSubFilter            run() on a worker thread; dispatchGPU(f) marshals f onto the thread that calls
//                             step() (the GL thread in the editor) and blocks until it has run
//
// Project here is only what the simulation filters read and write: canvas geometry and the host copies of the
// `img` terrain layer and the scratch1/scratch2 textures (what Texture::downloadDataRAW / uploadData move).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

struct Project {
    int width = 0, height = 0;
    float coords[4] = { -1.5707963267948966f, 1.5707963267948966f, -3.14159265358979f, 3.14159265358979f };   // src/Project.h:97
    float circumference = 42000.0f;                                                                             // src/Project.h:88
    std::vector<float> terrain, scratch1, scratch2;      // W*H fp32, row 0 = latitude coords[0]
    std::vector<float> selection;                        // empty = everything selected (the reference's `sel` texture, 0..1 per texel)
    int finalized = 0;                                   // how often a filter has ended (Project::finalizeFilter, src/Project.cpp)
    int getWidth() const { return width; }
    int getHeight() const { return height; }
    void finalizeFilter() { finalized++; }
};

class SubFilter {
public:
    virtual ~SubFilter() {}
    virtual std::pair<bool, float> step(Project *p) = 0;
};

class AsyncSubFilter : public SubFilter {
public:
    // The reference's Cancel (src/filter/Filter.cpp:105-108) restores the backup but lets the worker run on -- 10000 outer ocean
    // steps, half an hour at 16384x8192 -- and ~AsyncSubFilter would join it.  Here cancel() raises a flag the adapters' run()
    // loops read once per outer step / round / frame (cancelled()), a worker parked in dispatchGPU is released without its closure
    // being run, and the destructor cancels before it joins.
    // (the join must come BEFORE the derived adapter's members go -- until it the worker is inside the derived run(),
    // reading them.  Every adapter's destructor therefore calls stop() first; the call here is the safety net for a filter that
    // has no state of its own.)
    ~AsyncSubFilter() override { stop(); }
    void cancel()
    {
        cancel_requested.store(true);
        std::lock_guard<std::mutex> lk(gpu_mtx);
        cv.notify_all();
    }
    bool cancelled() const { return cancel_requested.load(); }

    std::pair<bool, float> step(Project *p) override
    {
        if (!started) {
            started = true;
            progress = { false, 0.0f };
            worker = std::thread([this] { run(); });
        }
        {
            std::unique_lock<std::mutex> lk(gpu_mtx);
            if (pending) {                 // run the closure the worker posted, on this (the "GL") thread
                closure(p);
                pending = false;
                cv.notify_all();
            }
        }
        auto pr = getProgress();
        if (pr.first && worker.joinable()) worker.join();
        return pr;
    }

protected:
    virtual void run() = 0;
    // cancel and join: the first statement of every derived destructor
    void stop() { cancel(); if (worker.joinable()) worker.join(); }

    std::pair<bool, float> getProgress()
    {
        std::lock_guard<std::mutex> lk(progress_mtx);
        return progress;
    }
    void setProgress(std::pair<bool, float> p)
    {
        std::lock_guard<std::mutex> lk(progress_mtx);
        progress = p;
    }
    // blocks the worker until a step() call has executed f (or the filter has been cancelled: f is then dropped)
    void dispatchGPU(std::function<void(Project *)> f)
    {
        std::unique_lock<std::mutex> lk(gpu_mtx);
        if (cancel_requested.load()) return;
        closure = std::move(f);
        pending = true;
        cv.wait(lk, [this] { return !pending || cancel_requested.load(); });
        pending = false;
    }

private:
    bool started = false;
    std::pair<bool, float> progress{ false, 0.0f };
    std::mutex progress_mtx, gpu_mtx;
    std::condition_variable cv;
    std::function<void(Project *)> closure;
    bool pending = false;
    std::atomic<bool> cancel_requested{false};
    std::thread worker;
};

// ---- Filter / BackupFilter / ProgressFilter (reference src/filter/Filter.h:23-56,117-130, src/filter/Filter.cpp:12-23,70-115) --------
// What Project keeps while a tool runs.  BackupFilter copies the target layer when it is made (Filter.cpp:12-23: the copy_img draw
// into `tmp`), restoreBackup() puts it back (:70-76), restoreUnselected() keeps the result only where the selection says so (:52-68:
// fc = s * result + (1 - s) * backup).  ProgressFilter::run() is what the editor calls once per frame (:97-115): one step of the
// sub-filter, the progress bar, and -- when Cancel has set `aborting` (:85-88) -- the backup restored and the filter finalized.  The
// reference lets the worker of an AsyncSubFilter run on after that (it is joined when the filter is destroyed, :139-141, after up to
// 10000 outer steps); here abort() also cancels it, and its parked closure -- the upload of a result nobody wants any more -- is dropped.
// `target` picks the layer as the reference's std::function<Texture *(Project *)> does.
class Filter {
public:
    virtual ~Filter() = default;
    virtual void run(Project *p) = 0;
    virtual bool isFinished() = 0;
};

class BackupFilter : public Filter {
public:
    BackupFilter(Project *p, std::function<std::vector<float> *(Project *)> target) : p(p), target(std::move(target)) { tmp = *this->target(p); }
    void restoreBackup() { *target(p) = tmp; }
    void restoreUnselected()
    {
        std::vector<float> &t = *target(p);
        if (p->selection.size() != t.size() || tmp.size() != t.size()) return;           // no selection: everything is selected
        for (size_t i = 0; i < t.size(); i++) { const float s = p->selection[i]; t[i] = s * t[i] + (1.0f - s) * tmp[i]; }
    }
    const std::vector<float> &backup() const { return tmp; }
protected:
    Project *p;
    std::vector<float> tmp;
    std::function<std::vector<float> *(Project *)> target;
};

class ProgressFilter : public BackupFilter {
public:
    ProgressFilter(Project *p, std::function<std::vector<float> *(Project *)> target, std::unique_ptr<SubFilter> subfilter)
        : BackupFilter(p, std::move(target)), subFilter(std::move(subfilter)) {}
    void abort() { aborting = true; }                     // the modal's Cancel button (Filter.cpp:85-88)
    float getProgress() const { return progress; }
    bool isFinished() override { return finished; }
    bool wasAborted() const { return aborted; }
    void run(Project *p) override
    {
        if (finished) return;
        if (aborting) {
            // (before the step: a closure the worker has parked -- the upload of its result -- must not run after the restore)
            if (auto *a = dynamic_cast<AsyncSubFilter *>(subFilter.get())) a->cancel();
            restoreBackup();
            finished = aborted = true;
            p->finalizeFilter();
            return;
        }
        auto pr = subFilter->step(p);
        finished = pr.first;
        progress = pr.second;
        if (finished) p->finalizeFilter();
    }
private:
    float progress = 0.0f;
    bool aborting = false, finished = false, aborted = false;
    std::unique_ptr<SubFilter> subFilter;
};

// Texture::downloadDataRAW hands over exactly width*height floats (reference src/Texture.cpp:56-67); a layer vector of
// another size would make the C ABI read past its end.
inline bool layer_matches(const std::vector<float> &layer, int W, int H)
{
    return W > 0 && H > 0 && layer.size() == (size_t)W * (size_t)H;
}
