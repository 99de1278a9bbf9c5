// render_task ROOT_DIR TASK_JSON OUT_FILE [seed]
// What the reference worker does with one task (worker/src/main.rs:75-100 without the
// sockets): parse the task, load its scene from ROOT_DIR, render on GPU 0, write the
// RenderedImage bytes (u32 LE width, u32 LE height, f32 LE rgb) to OUT_FILE.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "pathtracing_amd.hpp"

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s ROOT_DIR TASK_JSON OUT_FILE [seed]\n", argv[0]);
        return 2;
    }
    try {
        std::ifstream in(argv[2]);
        if (!in) throw std::runtime_error(std::string("cannot read ") + argv[2]);
        std::stringstream text;
        text << in.rdbuf();
        pt::RenderTask task = pt::RenderTask::from_json(text.str());
        PtRenderOpts opts{};
        opts.seed = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 0;
        PtRenderStats stats{};
        pt::Worker worker(argv[1]);
        pt::RenderedImage image = worker.render(task, &opts, &stats);
        std::vector<uint8_t> bytes = image.to_bytes();
        std::ofstream out(argv[3], std::ios::binary);
        out.write(reinterpret_cast<const char*>(bytes.data()), (std::streamsize)bytes.size());
        if (!out) throw std::runtime_error(std::string("cannot write ") + argv[3]);
        std::printf("%ux%u, %llu paths, %llu rays, %.2f ms on the device\n", image.width, image.height,
                    (unsigned long long)stats.paths, (unsigned long long)stats.rays, stats.render_ms);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "render_task: %s\n", e.what());
        return 1;
    }
    return 0;
}
