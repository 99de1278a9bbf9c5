// worker_front_end discovery [PORT]                      the UDP reply of main.rs:95-113, as hex
// worker_front_end digest ROOT_DIR SCENE_PATH            the scene_md5 the client names a scene's bucket with
// worker_front_end serve STORE_ROOT TASK_JSON OUT_FILE [seed]
//                                                        one round of connection_loop (main.rs:68-93): the text frame in
//                                                        TASK_JSON, its scene from STORE_ROOT/<scene_md5>/, the reply's
//                                                        bytes to OUT_FILE
// worker_front_end listen STORE_ROOT [WS_PORT UDP_PORT [SECONDS]]
//                                                        the worker binary (main.rs:28-48): WebSocket listener (30000) and
//                                                        discovery responder (40000) over STORE_ROOT's buckets, for
//                                                        SECONDS (default: until killed); prints the ports it bound
// worker_front_end multi ROOT_DIR TASK_JSON OUT_FILE N [seed]
//                                                        the task in N bands of rows on N renderers of GPU 0, summed
//                                                        (the path a worker with several GPUs takes, here on one)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <thread>
#include <chrono>

#include "pathtracing_amd.hpp"

static std::string slurp(const char* path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error(std::string("cannot read ") + path);
    std::stringstream text;
    text << in.rdbuf();
    return text.str();
}

static void spill(const char* path, const std::vector<uint8_t>& bytes) {
    std::ofstream out(path, std::ios::binary);
    out.write(reinterpret_cast<const char*>(bytes.data()), (std::streamsize)bytes.size());
    if (!out) throw std::runtime_error(std::string("cannot write ") + path);
}

int main(int argc, char** argv) {
    try {
        if (argc >= 2 && !std::strcmp(argv[1], "discovery")) {
            if (!pt::is_discovery_request(nullptr, 0)) throw std::runtime_error("the empty datagram is a request");
            for (uint8_t b : pt::discovery_response(argc > 2 ? (uint16_t)std::atoi(argv[2]) : 30000)) std::printf("%02x", b);
            std::printf("\n");
            return 0;
        }
        if (argc == 4 && !std::strcmp(argv[1], "digest")) {
            std::printf("%s\n", pt::scene_md5(argv[2], argv[3]).c_str());
            return 0;
        }
        if (argc >= 5 && !std::strcmp(argv[1], "serve")) {
            PtRenderOpts opts{};
            opts.seed = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 0;
            pt::WorkerFrontEnd worker(argv[2]);
            std::vector<uint8_t> reply = worker.handle_message(slurp(argv[3]), &opts);
            spill(argv[4], reply);
            std::printf("%zu bytes, %u scene(s) cached\n", reply.size(), worker.cached_scenes());
            return 0;
        }
        if (argc >= 3 && !std::strcmp(argv[1], "listen")) {
            pt::WorkerFrontEnd worker(argv[2]);
            pt::Server server(worker, argc > 4 ? (uint16_t)std::atoi(argv[3]) : 30000, argc > 4 ? (uint16_t)std::atoi(argv[4]) : 40000);
            std::printf("Listening on: 0.0.0.0:%u (discovery on %u)\n", server.websocket_port(), server.udp_port());
            std::fflush(stdout);
            const double seconds = argc > 5 ? std::atof(argv[5]) : -1.0;
            for (double waited = 0.0; seconds < 0.0 || waited < seconds; waited += 0.1)
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            PtServerStats st = server.stats();
            std::printf("%llu connection(s), %llu message(s) answered, %llu failed, %llu datagram(s)\n",
                        (unsigned long long)st.connections, (unsigned long long)st.messages_answered,
                        (unsigned long long)st.messages_failed, (unsigned long long)st.datagrams_answered);
            return 0;
        }
        if (argc >= 6 && !std::strcmp(argv[1], "multi")) {
            pt::RenderTask task = pt::RenderTask::from_json(slurp(argv[3]));
            PtRenderOpts opts{};
            opts.seed = argc > 6 ? std::strtoull(argv[6], nullptr, 10) : 0;
            auto scene = pt::Scene::load(argv[2], task.scene);
            pt::MultiRenderer renderers = pt::MultiRenderer::init(scene, std::vector<int>((size_t)std::atoi(argv[5]), 0));
            PtRenderStats stats{};
            pt::RenderedImage image = renderers.render(task, PT_SHARD_ROWS, &opts, &stats);
            spill(argv[4], image.to_bytes());
            std::printf("%ux%u, %llu rays\n", image.width, image.height, (unsigned long long)stats.rays);
            return 0;
        }
        std::fprintf(stderr, "usage: %s discovery [PORT] | digest ROOT SCENE | serve STORE TASK OUT [seed] | listen STORE [WS UDP [SECONDS]] | multi ROOT TASK OUT N [seed]\n", argv[0]);
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "worker_front_end: %s\n", e.what());
        return 1;
    }
}
