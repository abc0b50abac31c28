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
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

struct Project {
    int width = 0, height = 0;
    float coords[4] = { -1.5707963267948966f, 1.5707963267948966f, -3.14159265358979f, 3.14159265358979f };   // src/Project.h:97
    float circumference = 42000.0f;                                                                             // src/Project.h:88
    std::vector<float> terrain, scratch1, scratch2;      // W*H fp32, row 0 = latitude coords[0]
    int getWidth() const { return width; }
    int getHeight() const { return height; }
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
    ~AsyncSubFilter() override { cancel(); if (worker.joinable()) worker.join(); }
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

// Texture::downloadDataRAW hands over exactly width*height floats (reference src/Texture.cpp:56-67); a layer vector of
// another size would make the C ABI read past its end.
inline bool layer_matches(const std::vector<float> &layer, int W, int H)
{
    return W > 0 && H > 0 && layer.size() == (size_t)W * (size_t)H;
}
