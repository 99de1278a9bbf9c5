#include "listener.hpp"

#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <memory>

#include "front_end.hpp"
#include "host_scene.hpp"

namespace pt {

// ------------------------------------------------------------------ SHA-1, base64
void sha1(const uint8_t* data, size_t size, uint8_t out[20]) {
    uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
    auto rotl = [](uint32_t x, int n) { return (x << n) | (x >> (32 - n)); };
    auto block = [&](const uint8_t* p) {
        uint32_t w[80];
        for (int i = 0; i < 16; i++)
            w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 80; i++) w[i] = rotl(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
        for (int i = 0; i < 80; i++) {
            uint32_t f, k;
            if (i < 20) {
                f = (b & c) | (~b & d);
                k = 0x5A827999u;
            } else if (i < 40) {
                f = b ^ c ^ d;
                k = 0x6ED9EBA1u;
            } else if (i < 60) {
                f = (b & c) | (b & d) | (c & d);
                k = 0x8F1BBCDCu;
            } else {
                f = b ^ c ^ d;
                k = 0xCA62C1D6u;
            }
            uint32_t t = rotl(a, 5) + f + e + k + w[i];
            e = d;
            d = c;
            c = rotl(b, 30);
            b = a;
            a = t;
        }
        h[0] += a;
        h[1] += b;
        h[2] += c;
        h[3] += d;
        h[4] += e;
    };
    size_t full = size / 64;
    for (size_t i = 0; i < full; i++) block(data + 64 * i);
    uint8_t tail[128] = {0};
    size_t rest = size - 64 * full;
    memcpy(tail, data + 64 * full, rest);
    tail[rest] = 0x80;
    size_t padded = rest + 1 + 8 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)size * 8;
    for (int i = 0; i < 8; i++) tail[padded - 1 - i] = (uint8_t)(bits >> (8 * i));
    block(tail);
    if (padded == 128) block(tail + 64);
    for (int i = 0; i < 5; i++)
        for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(h[i] >> (24 - 8 * k));
}

std::string base64(const uint8_t* data, size_t size) {
    static const char* kAlphabet = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string out;
    out.reserve((size + 2) / 3 * 4);
    for (size_t i = 0; i < size; i += 3) {
        uint32_t v = (uint32_t)data[i] << 16;
        if (i + 1 < size) v |= (uint32_t)data[i + 1] << 8;
        if (i + 2 < size) v |= data[i + 2];
        out.push_back(kAlphabet[(v >> 18) & 63]);
        out.push_back(kAlphabet[(v >> 12) & 63]);
        out.push_back(i + 1 < size ? kAlphabet[(v >> 6) & 63] : '=');
        out.push_back(i + 2 < size ? kAlphabet[v & 63] : '=');
    }
    return out;
}

std::string websocket_accept(const std::string& key) {
    const std::string joined = key + "258EAFA5-E914-47DA-95CA-C5AB0DC85B11";
    uint8_t digest[20];
    sha1(reinterpret_cast<const uint8_t*>(joined.data()), joined.size(), digest);
    return base64(digest, sizeof(digest));
}

namespace {

constexpr int kPollMs = 100;                  // how often a blocked thread looks at the stop flag
constexpr size_t kMaxHandshake = 16 * 1024;   // bytes of HTTP request accepted before the blank line
constexpr uint64_t kMaxMessage = 64ull << 20; // tungstenite's default max_message_size

std::string lower(std::string s) {
    for (char& c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t\r");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

// A socket read through a buffer, giving up when the server stops.
struct Reader {
    int fd;
    const std::atomic<bool>& stopping;
    std::vector<uint8_t> buf;
    size_t pos = 0;

    // Waits for more bytes; false on end of stream, error or stop.
    bool fill() {
        for (;;) {
            if (stopping.load()) return false;
            pollfd p{fd, POLLIN, 0};
            int r = poll(&p, 1, kPollMs);
            if (r < 0 && errno == EINTR) continue;
            if (r < 0) return false;
            if (r == 0) continue;
            uint8_t chunk[65536];
            ssize_t n = recv(fd, chunk, sizeof(chunk), 0);
            if (n < 0 && (errno == EINTR || errno == EAGAIN)) continue;
            if (n <= 0) return false;
            if (pos > 0 && pos == buf.size()) {
                buf.clear();
                pos = 0;
            }
            buf.insert(buf.end(), chunk, chunk + n);
            return true;
        }
    }
    bool read(uint8_t* out, size_t n) {
        while (buf.size() - pos < n)
            if (!fill()) return false;
        memcpy(out, buf.data() + pos, n);
        pos += n;
        return true;
    }
};

bool send_all(int fd, const uint8_t* data, size_t size) {
    while (size > 0) {
        ssize_t n = send(fd, data, size, MSG_NOSIGNAL);
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) return false;
        data += n;
        size -= (size_t)n;
    }
    return true;
}

bool send_frame(int fd, uint8_t opcode, const uint8_t* payload, uint64_t size) {
    uint8_t head[10];
    size_t n = 0;
    head[n++] = (uint8_t)(0x80u | opcode); // FIN; frames from the server are not masked (RFC 6455 5.1)
    if (size < 126) {
        head[n++] = (uint8_t)size;
    } else if (size <= 0xFFFF) {
        head[n++] = 126;
        head[n++] = (uint8_t)(size >> 8);
        head[n++] = (uint8_t)size;
    } else {
        head[n++] = 127;
        for (int i = 7; i >= 0; i--) head[n++] = (uint8_t)(size >> (8 * i));
    }
    return send_all(fd, head, n) && (size == 0 || send_all(fd, payload, (size_t)size));
}

void send_close(int fd, uint16_t code) {
    uint8_t p[2] = {(uint8_t)(code >> 8), (uint8_t)code};
    send_frame(fd, 0x8, p, 2);
}

int bound_socket(int type, const std::string& addr, uint16_t port, uint16_t* bound_port) {
    int fd = socket(AF_INET, type, 0);
    if (fd < 0) throw Error(PT_ERR_IO, std::string("socket: ") + strerror(errno));
    int one = 1;
    setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    if (type == SOCK_DGRAM) setsockopt(fd, SOL_SOCKET, SO_BROADCAST, &one, sizeof(one)); // main.rs:99
    sockaddr_in sa;
    memset(&sa, 0, sizeof(sa));
    sa.sin_family = AF_INET;
    sa.sin_port = htons(port);
    if (inet_pton(AF_INET, addr.c_str(), &sa.sin_addr) != 1) {
        close(fd);
        throw Error(PT_ERR_INVALID_ARGUMENT, "bind address is not an IPv4 address: " + addr);
    }
    if (bind(fd, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) != 0) {
        std::string why = strerror(errno);
        close(fd);
        throw Error(PT_ERR_IO, "Failed to bind " + addr + ":" + std::to_string(port) + ": " + why);
    }
    socklen_t len = sizeof(sa);
    getsockname(fd, reinterpret_cast<sockaddr*>(&sa), &len);
    *bound_port = ntohs(sa.sin_port);
    return fd;
}

} // namespace

// ------------------------------------------------------------------------ server
Server::Server(const std::string& bind_addr, uint16_t websocket_port, uint16_t udp_port, MessageHandler handler)
    : handler_(std::move(handler)) {
    listen_fd_ = bound_socket(SOCK_STREAM, bind_addr, websocket_port, &ws_port_);
    if (listen(listen_fd_, 16) != 0) {
        std::string why = strerror(errno);
        close(listen_fd_);
        throw Error(PT_ERR_IO, "listen: " + why);
    }
    try {
        udp_fd_ = bound_socket(SOCK_DGRAM, bind_addr, udp_port, &udp_port_);
    } catch (...) {
        close(listen_fd_);
        throw;
    }
    accept_thread_ = std::thread([this] { accept_loop(); });
    udp_thread_ = std::thread([this] { discovery_loop(); });
}

Server::~Server() { stop(); }

void Server::stop() {
    if (stopping_.exchange(true)) return;
    if (accept_thread_.joinable()) accept_thread_.join();
    if (udp_thread_.joinable()) udp_thread_.join();
    std::vector<Connection> threads;
    {
        std::lock_guard<std::mutex> lock(mu_);
        threads.swap(connection_threads_);
    }
    for (Connection& c : threads)
        if (c.thread.joinable()) c.thread.join();
    if (listen_fd_ >= 0) close(listen_fd_);
    if (udp_fd_ >= 0) close(udp_fd_);
    listen_fd_ = udp_fd_ = -1;
}

ServerCounters Server::counters() const {
    std::lock_guard<std::mutex> lock(mu_);
    return counters_;
}

std::string Server::last_error() const {
    std::lock_guard<std::mutex> lock(mu_);
    return last_error_;
}

void Server::fail(const std::string& why) {
    std::lock_guard<std::mutex> lock(mu_);
    last_error_ = why;
}

// main.rs:95-113: every datagram that decodes as a discovery::Request -- a struct without fields, so every
// datagram -- is answered with a Response naming the WebSocket port.
void Server::discovery_loop() {
    while (!stopping_.load()) {
        pollfd p{udp_fd_, POLLIN, 0};
        int r = poll(&p, 1, kPollMs);
        if (r <= 0) continue;
        uint8_t buffer[1024]; // main.rs:103
        sockaddr_in from;
        socklen_t from_len = sizeof(from);
        ssize_t n = recvfrom(udp_fd_, buffer, sizeof(buffer), 0, reinterpret_cast<sockaddr*>(&from), &from_len);
        if (n < 0) continue;
        std::vector<uint8_t> reply = discovery_response_encode(ws_port_);
        if (sendto(udp_fd_, reply.data(), reply.size(), 0, reinterpret_cast<sockaddr*>(&from), from_len) >= 0) {
            std::lock_guard<std::mutex> lock(mu_);
            counters_.datagrams_answered++;
        }
    }
}

// main.rs:43-47: one task per accepted connection.
void Server::accept_loop() {
    while (!stopping_.load()) {
        pollfd p{listen_fd_, POLLIN, 0};
        int r = poll(&p, 1, kPollMs);
        if (r <= 0) continue;
        int fd = accept(listen_fd_, nullptr, nullptr);
        if (fd < 0) continue;
        int one = 1;
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        timeval tv{30, 0}; // a peer that stops reading must not hold a thread for ever
        setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
        std::lock_guard<std::mutex> lock(mu_);
        counters_.connections++;
        for (size_t i = 0; i < connection_threads_.size();) { // connections that have ended
            if (connection_threads_[i].done->load()) {
                connection_threads_[i].thread.join();
                connection_threads_.erase(connection_threads_.begin() + (long)i);
            } else {
                i++;
            }
        }
        auto done = std::make_shared<std::atomic<bool>>(false);
        connection_threads_.push_back({std::thread([this, fd, done] {
                                           connection(fd);
                                           close(fd);
                                           done->store(true);
                                       }),
                                       done});
    }
}

// handle_connection + connection_loop, main.rs:50-93.
void Server::connection(int fd) {
    Reader in{fd, stopping_, {}, 0};
    // ---- the HTTP upgrade (tokio_tungstenite::accept_async)
    std::string request;
    for (;;) {
        size_t have = in.buf.size() - in.pos;
        const uint8_t* b = in.buf.data() + in.pos;
        const uint8_t kEnd[4] = {'\r', '\n', '\r', '\n'};
        const uint8_t* e = std::search(b, b + have, kEnd, kEnd + 4);
        if (e != b + have) {
            request.assign(reinterpret_cast<const char*>(b), (size_t)(e - b));
            in.pos += (size_t)(e - b) + 4;
            break;
        }
        if (have > kMaxHandshake || !in.fill()) return fail("Error during the websocket handshake occurred: incomplete request");
    }
    std::string key, upgrade, connection_header, version;
    size_t line_end = request.find("\r\n");
    const std::string first = request.substr(0, line_end);
    size_t at = line_end == std::string::npos ? request.size() : line_end + 2;
    while (at < request.size()) {
        size_t e = request.find("\r\n", at);
        if (e == std::string::npos) e = request.size();
        std::string line = request.substr(at, e - at);
        at = e + 2;
        size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        std::string name = lower(trim(line.substr(0, colon))), value = trim(line.substr(colon + 1));
        if (name == "sec-websocket-key") key = value;
        else if (name == "upgrade") upgrade = lower(value);
        else if (name == "connection") connection_header = lower(value);
        else if (name == "sec-websocket-version") version = value;
    }
    const bool get = first.compare(0, 4, "GET ") == 0 && first.find(" HTTP/1.1") != std::string::npos;
    if (!get || key.empty() || upgrade.find("websocket") == std::string::npos ||
        connection_header.find("upgrade") == std::string::npos || version != "13") {
        static const char kBad[] = "HTTP/1.1 400 Bad Request\r\nConnection: close\r\nContent-Length: 0\r\n\r\n";
        send_all(fd, reinterpret_cast<const uint8_t*>(kBad), sizeof(kBad) - 1);
        return fail("Error during the websocket handshake occurred: not a WebSocket upgrade request");
    }
    {
        std::string ok = "HTTP/1.1 101 Switching Protocols\r\nConnection: Upgrade\r\nUpgrade: websocket\r\nSec-WebSocket-Accept: " +
                         websocket_accept(key) + "\r\n\r\n";
        if (!send_all(fd, reinterpret_cast<const uint8_t*>(ok.data()), ok.size())) return;
        std::lock_guard<std::mutex> lock(mu_);
        counters_.handshakes++;
    }
    // ---- connection_loop until a round fails
    std::vector<uint8_t> message, reply;
    for (;;) {
        // one message: frames up to the one with FIN, control frames in between answered on the spot
        message.clear();
        uint8_t kind = 0; // opcode of the message being assembled, 0 = none yet
        bool complete = false;
        std::string error;
        while (!complete) {
            uint8_t h[2];
            if (!in.read(h, 2)) return; // the peer went away (main.rs:72: the reference's task panics here)
            const bool fin = h[0] & 0x80u;
            const uint8_t opcode = h[0] & 0x0Fu;
            uint64_t len = h[1] & 0x7Fu;
            if (h[0] & 0x70u) error = "reserved bits set";
            if (!(h[1] & 0x80u)) error = "frame from the client is not masked";
            if (len == 126) {
                uint8_t e[2];
                if (!in.read(e, 2)) return;
                len = ((uint64_t)e[0] << 8) | e[1];
            } else if (len == 127) {
                uint8_t e[8];
                if (!in.read(e, 8)) return;
                len = 0;
                for (int i = 0; i < 8; i++) len = (len << 8) | e[i];
            }
            if (opcode >= 0x8 && (!fin || len > 125)) error = "fragmented or oversized control frame";
            if (len > kMaxMessage || message.size() + len > kMaxMessage) error = "message too long";
            if (!error.empty()) {
                send_close(fd, error == "message too long" ? 1009 : 1002);
                break;
            }
            uint8_t mask[4];
            if (!in.read(mask, 4)) return;
            const size_t before = message.size();
            std::vector<uint8_t> control;
            std::vector<uint8_t>& into = opcode >= 0x8 ? control : message;
            const size_t off = opcode >= 0x8 ? 0 : before;
            into.resize(off + (size_t)len);
            if (len && !in.read(into.data() + off, (size_t)len)) return;
            for (size_t i = 0; i < (size_t)len; i++) into[off + i] ^= mask[i & 3];
            if (opcode == 0x8) { // close: answer in kind, done
                send_frame(fd, 0x8, control.data(), std::min<size_t>(control.size(), 2));
                return;
            }
            if (opcode == 0x9) { // ping: tungstenite answers it before the application sees anything
                if (!send_frame(fd, 0xA, control.data(), control.size())) return;
                continue;
            }
            if (opcode == 0xA) continue; // unsolicited pong
            if (opcode == 0x0) {
                if (kind == 0) error = "continuation frame without a message";
            } else if (opcode == 0x1 || opcode == 0x2) {
                if (kind != 0) error = "new message inside a fragmented one";
                kind = opcode;
            } else {
                error = "unknown opcode";
            }
            if (!error.empty()) {
                send_close(fd, 1002);
                break;
            }
            complete = fin;
        }
        bool ok = error.empty();
        if (ok && kind != 0x1) { // main.rs:74-76
            error = "Unexpected message format";
            send_close(fd, 1003);
            ok = false;
        }
        if (ok) {
            reply.clear();
            const std::string text(reinterpret_cast<const char*>(message.data()), message.size());
            ok = handler_(text, &reply, &error); // main.rs:77-82
            if (!ok) send_close(fd, 1011);
        }
        if (ok && !send_frame(fd, 0x2, reply.data(), reply.size())) { // main.rs:84-90
            error = "Failed to send render result";
            ok = false;
        }
        std::lock_guard<std::mutex> lock(mu_);
        if (ok) {
            counters_.messages_answered++;
        } else {
            // main.rs:60-65: "Error during message exchange: {}", and the connection is dropped
            counters_.messages_failed++;
            last_error_ = error;
            return;
        }
    }
}

} // namespace pt
