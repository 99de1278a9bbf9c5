// The sockets of the reference worker's front end (worker/src/main.rs:36-113): the TCP listener
// whose connections are upgraded to WebSockets and then carry one RenderTask text frame in and one
// RenderedImage binary frame out per round, and the UDP socket that answers discovery broadcasts.
// Plain POSIX sockets and threads; the WebSocket side is the part of RFC 6455 a tungstenite client
// exercises (HTTP/1.1 upgrade, masked client frames, fragmented messages, close).
#pragma once
#include <atomic>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace pt {

// SHA-1 (RFC 3174) and base64 (RFC 4648) as the WebSocket handshake needs them.
void sha1(const uint8_t* data, size_t size, uint8_t out[20]);
std::string base64(const uint8_t* data, size_t size);
// Sec-WebSocket-Accept for a Sec-WebSocket-Key (RFC 6455 section 4.2.2).
std::string websocket_accept(const std::string& key);

struct ServerCounters {
    uint64_t connections = 0;       // TCP connections accepted
    uint64_t handshakes = 0;        // of those, upgraded to a WebSocket
    uint64_t messages_answered = 0; // text frames answered with a binary frame
    uint64_t messages_failed = 0;   // rounds that ended the connection (main.rs:60-65)
    uint64_t datagrams_answered = 0;
};

// What one round of connection_loop does with a text frame: fills `reply` (the binary frame's payload) or
// returns false with a message; false ends the connection (main.rs:60-65).
using MessageHandler = std::function<bool(const std::string& text, std::vector<uint8_t>* reply, std::string* error)>;

class Server {
  public:
    // Binds both sockets (port 0 = any free port) on `bind_addr` and starts the accept loop and the
    // discovery responder.  Throws pt::Error(PT_ERR_IO) when a socket cannot be bound.
    Server(const std::string& bind_addr, uint16_t websocket_port, uint16_t udp_port, MessageHandler handler);
    ~Server(); // stop()
    void stop(); // closes the listeners, ends the connections, joins every thread
    uint16_t websocket_port() const { return ws_port_; }
    uint16_t udp_port() const { return udp_port_; }
    ServerCounters counters() const;
    std::string last_error() const;

  private:
    void accept_loop();
    void discovery_loop();
    void connection(int fd);
    void fail(const std::string& why);

    MessageHandler handler_;
    int listen_fd_ = -1, udp_fd_ = -1;
    uint16_t ws_port_ = 0, udp_port_ = 0;
    std::atomic<bool> stopping_{false};
    std::thread accept_thread_, udp_thread_;
    mutable std::mutex mu_; // counters, last error, connection threads
    ServerCounters counters_;
    std::string last_error_;
    struct Connection {
        std::thread thread;
        std::shared_ptr<std::atomic<bool>> done;
    };
    std::vector<Connection> connection_threads_;
};

} // namespace pt
