// context.hxx -- per-device execution context: stream, events, timer.
// API parity: include/gunrock/cuda/context.hxx:54-216 (reference):
// gcuda::standard_context_t {stream(), synchronize(), timer(), event(), props(),
// ordinal(), execution_policy()} and gcuda::multi_context_t {get_context(i),
// size(), enable_peer_access()}.
// MI355X additions: the context owns a small pinned mailbox through which
// operators read back sizes with a single stream sync (the reference copies
// through 1-element thrust vectors, advance/helpers.hxx:106-110).
#pragma once

#include <hip/hip_runtime.h>
#include <thrust/execution_policy.h>
#include <thrust/system/hip/execution_policy.h>

#include <memory>
#include <vector>

#include <gunrock/error.hxx>
#include <gunrock/util/timer.hxx>

namespace gunrock {
namespace gcuda {

typedef int device_id_t;
typedef hipStream_t stream_t;
typedef hipEvent_t event_t;
typedef hipDeviceProp_t device_properties_t;

class standard_context_t {
  device_properties_t _props;
  device_id_t _ordinal;
  stream_t _stream = nullptr;
  event_t _event = nullptr;
  bool _own_stream = true;
  util::timer_t _timer;
  int* _mailbox = nullptr;  // pinned host, 64 ints
  int* _mailbox_dev = nullptr;  // the same words as kernels address them (null: not mapped -- sizes come back by copy)
  void* _scratch[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // growable device scratch slots
  std::size_t _scratch_bytes[5] = {0, 0, 0, 0, 0};
  // state another layer ties to THIS context's lifetime (the pre-compiled engine keeps its
  // grx_context here: it references _stream and must die before it)
  void* _attached = nullptr;
  void (*_attached_free)(void*) = nullptr;

  void init() {
    error::throw_if_exception(hipSetDevice(_ordinal), "hipSetDevice");
    if (_own_stream)
      error::throw_if_exception(hipStreamCreateWithFlags(&_stream, hipStreamNonBlocking), "stream create");
    error::throw_if_exception(hipEventCreateWithFlags(&_event, hipEventDisableTiming), "event create");
    error::throw_if_exception(hipGetDeviceProperties(&_props, _ordinal), "device properties");
    error::throw_if_exception(hipHostMalloc(reinterpret_cast<void**>(&_mailbox), 64 * sizeof(int), hipHostMallocMapped),
                              "pinned mailbox");
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, _mailbox, 0) == hipSuccess) _mailbox_dev = reinterpret_cast<int*>(dev);
    (void)hipGetLastError();
  }

 public:
  standard_context_t(device_id_t device = 0) : _ordinal(device) { init(); }
  standard_context_t(hipStream_t stream, device_id_t device = 0)
      : _ordinal(device), _stream(stream), _own_stream(false) { init(); }
  standard_context_t(const standard_context_t&) = delete;
  ~standard_context_t() {
    if (_attached && _attached_free) _attached_free(_attached);
    if (_event) (void)hipEventDestroy(_event);
    if (_mailbox) (void)hipHostFree(_mailbox);
    for (auto* p : _scratch)
      if (p) (void)hipFree(p);
    if (_own_stream && _stream) (void)hipStreamDestroy(_stream);
  }

  const device_properties_t& props() const { return _props; }
  stream_t stream() { return _stream; }
  event_t event() { return _event; }
  util::timer_t& timer() { return _timer; }
  device_id_t ordinal() { return _ordinal; }
  int* mailbox() { return _mailbox; }
  void* attached() const { return _attached; }
  void attach(void* state, void (*free_fn)(void*)) {
    _attached = state;
    _attached_free = free_fn;
  }

  void synchronize() {
    error::throw_if_exception(_stream ? hipStreamSynchronize(_stream) : hipDeviceSynchronize(), "synchronize");
  }

  auto execution_policy() { return thrust::hip::par.on(this->stream()); }

  // Persistent device scratch (operators never hipMalloc per call; the reference
  // allocates a 1-element device_vector inside every block_mapped advance,
  // advance/block_mapped.hxx:244).
  template <typename type_t>
  type_t* scratch(int slot, std::size_t count) {
    const std::size_t need = count * sizeof(type_t);
    if (need > _scratch_bytes[slot]) {
      if (_scratch[slot]) {
        synchronize();
        error::throw_if_exception(hipFree(_scratch[slot]), "scratch free");
      }
      const std::size_t want = need + need / 4 + 256;
      error::throw_if_exception(hipMalloc(&_scratch[slot], want), "scratch alloc");
      _scratch_bytes[slot] = want;
    }
    return reinterpret_cast<type_t*>(_scratch[slot]);
  }

  // Words [32, 64) of the mailbox as a kernel writes them (a size a single-workgroup kernel computes goes straight to the
  // host: no copy behind it); wait_mailbox() = the stream has drained, the words are current.  Null when the mailbox is not
  // device-visible.
  int* mailbox_device(int slot) { return _mailbox_dev ? _mailbox_dev + 32 + slot : nullptr; }
  const int* wait_mailbox(int slot) {
    synchronize();
    return _mailbox + 32 + slot;
  }

  // Read `count` ints from device memory through the pinned mailbox (one sync).
  const int* read_back(const int* d_src, int count = 1) {
    error::throw_if_exception(
        hipMemcpyAsync(_mailbox, d_src, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost, _stream), "read_back");
    synchronize();
    return _mailbox;
  }
};

class multi_context_t {
 public:
  std::vector<standard_context_t*> contexts;
  std::vector<device_id_t> devices;
  static constexpr std::size_t MAX_NUMBER_OF_GPUS = 1024;

  multi_context_t(std::vector<device_id_t> _devices) : devices(_devices) {
    for (auto d : devices) contexts.push_back(new standard_context_t(d));
  }
  multi_context_t(std::vector<device_id_t> _devices, hipStream_t _stream) : devices(_devices) {
    for (auto d : devices) contexts.push_back(new standard_context_t(_stream, d));
  }
  multi_context_t(device_id_t _device) : devices(1, _device) {
    contexts.push_back(new standard_context_t(_device));
  }
  multi_context_t(device_id_t _device, hipStream_t _stream) : devices(1, _device) {
    contexts.push_back(new standard_context_t(_stream, _device));
  }
  multi_context_t(const multi_context_t&) = delete;
  ~multi_context_t() {
    for (auto* c : contexts) delete c;
  }

  standard_context_t* get_context(device_id_t device) { return contexts[(std::size_t)device]; }
  std::size_t size() { return contexts.size(); }

  void enable_peer_access() {
    const int n = (int)size();
    for (int i = 0; i < n; ++i) {
      (void)hipSetDevice(get_context(i)->ordinal());
      for (int j = 0; j < n; ++j) {
        if (i == j) continue;
        int ok = 0;
        (void)hipDeviceCanAccessPeer(&ok, get_context(i)->ordinal(), get_context(j)->ordinal());
        if (ok) (void)hipDeviceEnablePeerAccess(get_context(j)->ordinal(), 0);
      }
    }
  }
};

}  // namespace gcuda
}  // namespace gunrock
