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
    ~AsyncSubFilter() override { if (worker.joinable()) worker.join(); }

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
    // blocks the worker until a step() call has executed f
    void dispatchGPU(std::function<void(Project *)> f)
    {
        std::unique_lock<std::mutex> lk(gpu_mtx);
        closure = std::move(f);
        pending = true;
        cv.wait(lk, [this] { return !pending; });
    }

private:
    bool started = false;
    std::pair<bool, float> progress{ false, 0.0f };
    std::mutex progress_mtx, gpu_mtx;
    std::condition_variable cv;
    std::function<void(Project *)> closure;
    bool pending = false;
    std::thread worker;
};

// Texture::downloadDataRAW hands over exactly width*height floats (reference src/Texture.cpp:56-67); a layer vector of
// another size would make the C ABI read past its end.
inline bool layer_matches(const std::vector<float> &layer, int W, int H)
{
    return W > 0 && H > 0 && layer.size() == (size_t)W * (size_t)H;
}
