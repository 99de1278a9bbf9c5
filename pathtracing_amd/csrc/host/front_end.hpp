// The pieces of the reference worker's front end that are plain data handling, without
// any socket: the discovery messages' wire format, the scene digest that names a scene's
// file bucket, and MD5 itself.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "host_scene.hpp"

namespace pt {

// RFC 1321, as lower-case hex (what `format!("{:x}", md5::compute(..))` yields, client/src/scene.rs:23).
std::string md5_hex(const uint8_t* data, size_t size);

// The digest the client names a scene's GridFS bucket with (client/src/scene.rs:19-85): the MD5 of
// every file the scene refers to (the scene file itself, meshes, materials, images; paths as written),
// sorted by path, folded into one: acc = md5(acc + md5_i), starting from the empty string.  kd_tree
// files are left out: the client stops at them (`unimplemented!()`, scene.rs:59-61), and here they
// are a cache the worker may rewrite, not scene content.
std::string scene_md5(const FileStore& store, const std::string& scene_path);

// postcard 1.1 (Cargo.lock:3692) wire format of worker::discovery::{Request, Response}
// (worker/src/lib.rs:81-91): a struct is its fields in order, u16 is an LEB128 varint of at most
// three bytes, a struct without fields is no bytes at all.  Decoders ignore trailing bytes, as
// postcard::from_bytes does.
std::vector<uint8_t> discovery_response_encode(uint16_t websocket_port);
bool discovery_response_decode(const uint8_t* bytes, size_t len, uint16_t* websocket_port);

} // namespace pt
