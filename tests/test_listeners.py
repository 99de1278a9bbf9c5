"""The worker's sockets (worker/src/main.rs:36-113) on loopback with ephemeral ports: the UDP
discovery responder and the WebSocket listener, driven by a client written here from the
standard library (RFC 6455: upgrade request, masked frames).  Without a GPU: the handshake,
the discovery replies and every way a round can fail; with one: a RenderTask text frame in,
the RenderedImage binary frame out, equal to pt_worker_handle_message's bytes."""
import base64
import hashlib
import json
import os
import shutil
import socket
import struct
import time

import pytest

import pathtracing_amd as P
from pathtracing_amd import front_end as F
from tests import helpers as H

GUID = b"258EAFA5-E914-47DA-95CA-C5AB0DC85B11"


class WsClient:
    """What tungstenite's client does on the wire, as far as the reference client uses it."""

    def __init__(self, port, key=b"dGhlIHNhbXBsZSBub25jZQ==", extra_headers="", request=None):
        self.sock = socket.create_connection(("127.0.0.1", port), timeout=20)
        if request is None:
            request = (f"GET / HTTP/1.1\r\nHost: 127.0.0.1:{port}\r\nConnection: Upgrade\r\nUpgrade: websocket\r\n"
                       f"Sec-WebSocket-Version: 13\r\nSec-WebSocket-Key: {key.decode()}\r\n{extra_headers}\r\n").encode()
        self.sock.sendall(request)
        self.buf = b""
        head = self._until(b"\r\n\r\n")
        self.status_line, _, rest = head.partition(b"\r\n")
        self.headers = {}
        for line in rest.split(b"\r\n"):
            if b":" in line:
                k, v = line.split(b":", 1)
                self.headers[k.strip().lower()] = v.strip()

    def _until(self, marker):
        while marker not in self.buf:
            chunk = self.sock.recv(65536)
            if not chunk:
                out, self.buf = self.buf, b""
                return out
            self.buf += chunk
        out, _, self.buf = self.buf.partition(marker)
        return out

    def _exactly(self, n):
        while len(self.buf) < n:
            chunk = self.sock.recv(1 << 20)
            if not chunk:
                raise EOFError("connection closed")
            self.buf += chunk
        out, self.buf = self.buf[:n], self.buf[n:]
        return out

    def send_frame(self, opcode, payload, fin=True, masked=True):
        head = bytes([(0x80 if fin else 0) | opcode])
        n = len(payload)
        m = 0x80 if masked else 0
        if n < 126:
            head += bytes([m | n])
        elif n <= 0xFFFF:
            head += bytes([m | 126]) + struct.pack(">H", n)
        else:
            head += bytes([m | 127]) + struct.pack(">Q", n)
        if masked:
            key = b"\x12\x34\x56\x78"
            payload = bytes(b ^ key[i & 3] for i, b in enumerate(payload))
            head += key
        self.sock.sendall(head + payload)

    def recv_frame(self):
        """(opcode, payload) of the next frame from the server, which must be unmasked and final."""
        b0, b1 = self._exactly(2)
        assert b0 & 0x80 and not b1 & 0x80, "server frames are final and unmasked"
        n = b1 & 0x7F
        if n == 126:
            n = struct.unpack(">H", self._exactly(2))[0]
        elif n == 127:
            n = struct.unpack(">Q", self._exactly(8))[0]
        return b0 & 0x0F, self._exactly(n)

    def closed_by_peer(self):
        """True when the server has nothing more to say: a close frame and / or end of stream."""
        try:
            while True:
                op, _ = self.recv_frame()
                if op != 0x8:
                    return False
        except EOFError:
            return True

    def close(self):
        self.sock.close()


def wait_for(predicate, seconds=5.0):
    end = time.time() + seconds
    while time.time() < end:
        if predicate():
            return True
        time.sleep(0.02)
    return predicate()


@pytest.fixture
def served(tmp_path):
    worker = F.WorkerFrontEnd(str(tmp_path / "store"), 0)
    server = worker.serve(0, 0, "127.0.0.1")
    yield worker, server
    server.stop()
    worker.close()


def test_discovery_datagrams_are_answered_with_the_websocket_port(served):
    _, server = served
    assert server.websocket_port and server.udp_port and server.websocket_port != server.udp_port
    s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    s.settimeout(5)
    # postcard::to_allocvec(&Request {}) is no bytes; trailing bytes are ignored by from_bytes (main.rs:104-105)
    for datagram in (F.discovery_request_bytes(), b"\x00", b"anything at all", bytes(1024)):
        s.sendto(datagram, ("127.0.0.1", server.udp_port))
        reply, sender = s.recvfrom(64)
        assert sender[1] == server.udp_port
        assert reply == F.discovery_response_bytes(server.websocket_port)
        assert F.discovery_response_port(reply) == server.websocket_port
    s.close()
    # the responder counts a datagram after it has sent the reply
    assert wait_for(lambda: server.stats()["datagrams_answered"] == 4)


def test_websocket_upgrade_follows_rfc6455(served):
    _, server = served
    key = base64.b64encode(os.urandom(16))
    c = WsClient(server.websocket_port, key)
    assert c.status_line.startswith(b"HTTP/1.1 101")
    assert c.headers[b"upgrade"].lower() == b"websocket" and c.headers[b"connection"].lower() == b"upgrade"
    assert c.headers[b"sec-websocket-accept"] == base64.b64encode(hashlib.sha1(key + GUID).digest())
    # the RFC's own example (section 1.3)
    c2 = WsClient(server.websocket_port)
    assert c2.headers[b"sec-websocket-accept"] == b"s3pPLMBiTxaQ9kYGzzhZRbK+xOo="
    # header names in any case, a comma-separated Connection header (browsers, proxies)
    c3 = WsClient(server.websocket_port, request=(
        b"GET /x HTTP/1.1\r\nhost: h\r\nCONNECTION: keep-alive, Upgrade\r\nupgrade: WebSocket\r\n"
        b"sec-websocket-version: 13\r\nSEC-WEBSOCKET-KEY: dGhlIHNhbXBsZSBub25jZQ==\r\n\r\n"))
    assert c3.status_line.startswith(b"HTTP/1.1 101")
    for x in (c, c2, c3):
        x.close()
    assert wait_for(lambda: server.stats()["handshakes"] == 3)
    assert server.stats()["connections"] == 3


@pytest.mark.parametrize("request_bytes", [
    b"GET / HTTP/1.1\r\nHost: h\r\n\r\n",                                                       # a plain HTTP request
    b"GET / HTTP/1.1\r\nConnection: Upgrade\r\nUpgrade: websocket\r\nSec-WebSocket-Version: 13\r\n\r\n",  # no key
    b"GET / HTTP/1.1\r\nConnection: Upgrade\r\nUpgrade: websocket\r\nSec-WebSocket-Version: 8\r\nSec-WebSocket-Key: eA==\r\n\r\n",
    b"POST / HTTP/1.1\r\nConnection: Upgrade\r\nUpgrade: websocket\r\nSec-WebSocket-Version: 13\r\nSec-WebSocket-Key: eA==\r\n\r\n",
])
def test_requests_that_are_no_upgrade_are_refused(served, request_bytes):
    _, server = served
    c = WsClient(server.websocket_port, request=request_bytes)
    assert c.status_line.startswith(b"HTTP/1.1 400")
    assert c.sock.recv(16) == b""  # and the connection is closed
    c.close()
    st = server.stats()
    assert st["handshakes"] == 0 and "handshake" in st["last_error"]


def test_a_round_that_fails_ends_its_connection_and_no_other(served):
    worker, server = served
    # a text frame that is no RenderTask (main.rs:77-78), whole and in three fragments with a ping in between
    bad = json.dumps({"scene": "./scene.json", "config": {"trace_depth": 1, "iterations": 1}})
    c = WsClient(server.websocket_port)
    c.send_frame(0x1, bad.encode())
    assert c.closed_by_peer()
    assert wait_for(lambda: server.stats()["messages_failed"] == 1)
    assert "Failed to decode render task" in server.stats()["last_error"]
    c.close()

    c = WsClient(server.websocket_port)
    raw = b'{"scene": 5, "why": "' + b"x" * 70000 + b'"}'  # a 16-bit and a 64-bit length on the way
    c.send_frame(0x1, raw[:10], fin=False)
    c.send_frame(0x9, b"are you there")
    c.send_frame(0x0, raw[10:200], fin=False)
    c.send_frame(0x0, raw[200:], fin=True)
    op, payload = c.recv_frame()
    assert (op, payload) == (0xA, b"are you there")  # the pong; the message was assembled around it
    assert c.closed_by_peer()
    assert wait_for(lambda: server.stats()["messages_failed"] == 2)
    assert "Failed to decode render task" in server.stats()["last_error"]
    c.close()

    # a binary frame: "Unexpected message format" (main.rs:74-76)
    c = WsClient(server.websocket_port)
    c.send_frame(0x2, b"\x00\x01")
    assert c.closed_by_peer()
    assert wait_for(lambda: server.stats()["messages_failed"] == 3)
    assert server.stats()["last_error"] == "Unexpected message format"
    c.close()

    # a frame from the client that is not masked, a continuation out of nowhere, an unknown opcode: protocol errors
    for opcode, masked in ((0x1, False), (0x0, True), (0x3, True)):
        c = WsClient(server.websocket_port)
        c.send_frame(opcode, b"{}", masked=masked)
        assert c.closed_by_peer()
        c.close()
    assert wait_for(lambda: server.stats()["messages_failed"] == 6)

    # a close frame is answered with a close frame
    c = WsClient(server.websocket_port)
    c.send_frame(0x8, struct.pack(">H", 1000))
    assert c.recv_frame() == (0x8, struct.pack(">H", 1000))
    c.close()

    # a task whose bucket does not exist fails in Worker::render (lib.rs:32-41); the listener is still there afterwards
    with open(os.path.join(H.SCENE_DATA, "render_task.json")) as f:
        wire = json.load(f)
    wire["scene_md5"] = "0" * 32
    c = WsClient(server.websocket_port)
    c.send_frame(0x1, json.dumps(wire).encode())
    assert c.closed_by_peer()
    c.close()
    assert wait_for(lambda: server.stats()["messages_failed"] == 7)
    assert worker.cached_scene_count == 0 and server.stats()["messages_answered"] == 0
    c = WsClient(server.websocket_port)
    assert c.status_line.startswith(b"HTTP/1.1 101")
    c.close()


def test_ports_in_use_and_stop(tmp_path):
    worker = F.WorkerFrontEnd(str(tmp_path), 0)
    a = worker.serve(0, 0, "127.0.0.1")
    with pytest.raises(P._native.PathTracingError) as e:
        worker.serve(a.websocket_port, 0, "127.0.0.1")
    assert e.value.status == 3 and "Failed to bind" in e.value.message  # main.rs:40
    with pytest.raises(P._native.PathTracingError):
        worker.serve(0, 0, "not an address")
    # stop() ends open connections and frees both ports
    c = WsClient(a.websocket_port)
    ws, udp = a.websocket_port, a.udp_port
    t0 = time.time()
    a.stop()
    assert time.time() - t0 < 3.0
    assert c.closed_by_peer()
    c.close()
    b = worker.serve(ws, udp, "127.0.0.1")
    assert (b.websocket_port, b.udp_port) == (ws, udp)
    b.stop()
    worker.close()


@pytest.mark.gpu
def test_render_tasks_over_the_websocket(tmp_path):
    store = str(tmp_path / "store")
    os.makedirs(store)
    md5 = F.scene_md5(H.SCENE_DATA, "./simple_scene.json")
    shutil.copytree(H.SCENE_DATA, os.path.join(store, md5))
    with open(os.path.join(H.SCENE_DATA, "render_task.json")) as f:
        wire = json.load(f)
    wire["scene"] = "./simple_scene.json"
    wire["scene_md5"] = md5
    wire["camera"]["resolution"] = [96, 64]
    wire["config"] = {"trace_depth": 5, "iterations": 2}
    worker = F.WorkerFrontEnd(store, 0)
    with worker.serve(0, 0, "127.0.0.1") as server:
        # discovery first, as the client does (worker_pool.rs:137-157, 251-257): the reply names the port to connect to
        s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        s.settimeout(5)
        s.sendto(b"", ("127.0.0.1", server.udp_port))
        port = F.discovery_response_port(s.recvfrom(16)[0])
        s.close()
        c = WsClient(port)
        replies = []
        for spp in (2, 1):  # two rounds on one connection (worker_pool.rs:259-286 sends a task a second)
            wire["config"]["iterations"] = spp
            text = json.dumps(wire)
            c.send_frame(0x1, text.encode())
            op, payload = c.recv_frame()
            assert op == 0x2
            assert payload == worker.handle_message(text)  # the same task without the socket
            image = P.RenderedImage.from_bytes(payload).image
            assert image.shape == (64, 96, 3) and image.mean() > 0.01
            replies.append(payload)
        assert replies[0] != replies[1]
        # a frame that does not decode ends this connection; the worker and its cached scene live on
        c.send_frame(0x1, b"{}")
        assert c.closed_by_peer()
        c.close()
        c = WsClient(port)
        c.send_frame(0x1, json.dumps(wire).encode())
        assert c.recv_frame() == (0x2, replies[1])
        c.close()
        assert wait_for(lambda: server.stats()["messages_answered"] == 3) # counted after the reply has been sent
        assert server.stats()["messages_failed"] == 1 and worker.cached_scene_count == 1
    worker.close()


def test_listen_mode_of_the_native_example(tmp_path):
    """examples/worker_front_end listen: the worker binary's main() (main.rs:28-48) from compiled C++ through the header-only mirror."""
    import re
    import subprocess

    exe = os.path.join(H.ROOT, "examples", "worker_front_end")
    assert os.path.exists(exe), "examples/worker_front_end is built by __graft_entry__.build()"
    proc = subprocess.Popen([exe, "listen", str(tmp_path), "0", "0", "3"], stdout=subprocess.PIPE, text=True)
    try:
        m = re.match(r"Listening on: 0\.0\.0\.0:(\d+) \(discovery on (\d+)\)", proc.stdout.readline())
        assert m, "the example did not report its ports"
        ws, udp = int(m.group(1)), int(m.group(2))
        s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        s.settimeout(5)
        s.sendto(b"", ("127.0.0.1", udp))
        assert F.discovery_response_port(s.recvfrom(16)[0]) == ws
        s.close()
        c = WsClient(ws)
        assert c.status_line.startswith(b"HTTP/1.1 101")
        c.send_frame(0x2, b"not text")
        assert c.closed_by_peer()
        c.close()
        out, _ = proc.communicate(timeout=20)
        assert proc.returncode == 0 and "1 connection(s), 0 message(s) answered, 1 failed, 1 datagram(s)" in out
    finally:
        if proc.poll() is None:
            proc.kill()


def test_blocking_serve_until_the_flag_is_set(tmp_path):
    """pt_worker_serve: start_ws as one call that returns when *stop_flag becomes non-zero."""
    import ctypes as C
    import threading

    from pathtracing_amd import _native as N

    def free_port(kind):
        s = socket.socket(socket.AF_INET, kind)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        return port

    ws, udp = free_port(socket.SOCK_STREAM), free_port(socket.SOCK_DGRAM)
    worker = F.WorkerFrontEnd(str(tmp_path), 0)
    flag = C.c_int32(0)
    status = []
    t = threading.Thread(target=lambda: status.append(N.lib.pt_worker_serve(worker._handle, b"127.0.0.1", ws, udp, C.byref(flag))))
    t.start()
    s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    s.settimeout(0.5)
    reply = None
    for _ in range(20):  # until the sockets are up
        try:
            s.sendto(b"", ("127.0.0.1", udp))
            reply = s.recvfrom(16)[0]
            break
        except (socket.timeout, ConnectionError):
            time.sleep(0.1)
    s.close()
    assert reply is not None and F.discovery_response_port(reply) == ws
    c = WsClient(ws)
    assert c.status_line.startswith(b"HTTP/1.1 101")
    flag.value = 1
    t.join(5)
    assert not t.is_alive() and status == [0]
    assert c.closed_by_peer()
    c.close()
    worker.close()
