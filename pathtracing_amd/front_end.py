"""The reference worker's front end (``worker/src/main.rs:36-113``): the discovery messages
in their postcard wire format, the scene digest that names a scene's file bucket
(``client/src/scene.rs:19-85``), one round of ``connection_loop`` -- a ``RenderTask`` text
frame in, the ``RenderedImage`` binary frame out -- over a ``Worker`` whose ``FileStore``
buckets are directories named by ``scene_md5``, and the two sockets themselves (``Server``:
WebSocket listener + UDP discovery responder)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

from . import _native as N
from .api import RenderOpts

WEBSOCKET_PORT = 30000  # worker/src/main.rs:17
BROADCAST_PORT = 40000  # worker/src/main.rs:18


def discovery_request_bytes() -> bytes:
    """``postcard::to_allocvec(&discovery::Request {})``: no bytes at all."""
    n = C.c_size_t()
    buf = (C.c_uint8 * 4)()
    N.check(N.lib.pt_discovery_request_encode(buf, 4, C.byref(n)))
    return bytes(buf[: n.value])


def discovery_request_ok(datagram: bytes) -> bool:
    """``postcard::from_bytes::<discovery::Request>(datagram).is_ok()``."""
    raw = (C.c_uint8 * max(1, len(datagram))).from_buffer_copy(datagram or b"\0")
    return N.lib.pt_discovery_request_decode(raw, len(datagram)) == N.PT_OK


def discovery_response_bytes(websocket_port: int = WEBSOCKET_PORT) -> bytes:
    """``postcard::to_allocvec(&discovery::Response { websocket_port })``."""
    n = C.c_size_t()
    buf = (C.c_uint8 * 3)()
    N.check(N.lib.pt_discovery_response_encode(websocket_port, buf, 3, C.byref(n)))
    return bytes(buf[: n.value])


def discovery_response_port(datagram: bytes) -> int:
    """``postcard::from_bytes::<discovery::Response>(datagram)?.websocket_port``."""
    raw = (C.c_uint8 * max(1, len(datagram))).from_buffer_copy(datagram or b"\0")
    port = C.c_uint16()
    N.check(N.lib.pt_discovery_response_decode(raw, len(datagram), C.byref(port)))
    return int(port.value)


def scene_md5(root_dir: str, scene_path: str) -> str:
    out = C.create_string_buffer(33)
    N.check(N.lib.pt_scene_md5(root_dir.encode(), scene_path.encode(), out))
    return out.value.decode()


class WorkerFrontEnd:
    """``Worker`` behind ``connection_loop``: ``store_root/<scene_md5>/`` is the bucket of a scene."""

    def __init__(self, store_root: str, device: int = 0):
        h = C.c_void_p()
        N.check(N.lib.pt_worker_create(store_root.encode(), device, C.byref(h)))
        self._handle = h

    def handle_message(self, task_json: str, opts: Optional[RenderOpts] = None) -> bytes:
        data = C.POINTER(C.c_uint8)()
        size = C.c_uint64()
        o = opts.to_native() if opts is not None else None
        N.check(N.lib.pt_worker_handle_message(self._handle, task_json.encode("utf-8"), C.byref(o) if o is not None else None,
                                               C.byref(data), C.byref(size)))
        try:
            return C.string_at(data, size.value)
        finally:
            N.lib.pt_buffer_free(data)

    @property
    def cached_scene_count(self) -> int:
        return int(N.lib.pt_worker_cached_scene_count(self._handle))

    def serve(self, websocket_port: int = WEBSOCKET_PORT, udp_port: int = BROADCAST_PORT, bind_addr: str = "0.0.0.0") -> "Server":
        """``start_ws`` (``main.rs:36-48``): both sockets open and served by the library's threads until ``stop()``."""
        return Server(self, websocket_port, udp_port, bind_addr)

    def close(self) -> None:
        if self._handle:
            N.lib.pt_worker_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Server:
    """The worker's listeners (``worker/src/main.rs:36-113``) over a ``WorkerFrontEnd``; port 0 binds any free port."""

    def __init__(self, worker: WorkerFrontEnd, websocket_port: int = WEBSOCKET_PORT, udp_port: int = BROADCAST_PORT,
                 bind_addr: str = "0.0.0.0"):
        self._worker = worker  # the worker must outlive the server
        h = C.c_void_p()
        N.check(N.lib.pt_worker_serve_start(worker._handle, bind_addr.encode(), websocket_port, udp_port, C.byref(h)))
        self._handle = h
        ws, udp = C.c_uint16(), C.c_uint16()
        N.check(N.lib.pt_server_ports(h, C.byref(ws), C.byref(udp)))
        self.websocket_port, self.udp_port = int(ws.value), int(udp.value)

    def stats(self) -> dict:
        st = N.PtServerStats()
        err = C.create_string_buffer(512)
        N.check(N.lib.pt_server_stats(self._handle, C.byref(st), err, len(err)))
        out = {name: int(getattr(st, name)) for name, _ in N.PtServerStats._fields_}
        out["last_error"] = err.value.decode("utf-8", "replace")
        return out

    def stop(self) -> None:
        if self._handle:
            N.lib.pt_server_stop(self._handle)
            self._handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.stop()

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass
