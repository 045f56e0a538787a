// bfh_comm_*: RCCL communicator objects for the data-parallel paths (users sharded, item factors replicated;
// ALS rows sharded) -- SURVEY.md section 8(e).  The reference has no multi-device code.
#include "comm.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <thread>

namespace bfh {

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            x.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) throw Error(BFH_ERR_HIP, std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "?"));
        auto sym = [&](const char* name) {
            void* p = dlsym(x.lib, name);
            if (!p) throw Error(BFH_ERR_HIP, std::string("librccl: missing symbol ") + name);
            return p;
        };
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.CommCount = reinterpret_cast<decltype(x.CommCount)>(sym("ncclCommCount"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
        x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(sym("ncclBroadcast"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        x.GetVersion = reinterpret_cast<decltype(x.GetVersion)>(sym("ncclGetVersion"));
        return x;
    }();
    return r;
}

void check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw Error(BFH_ERR_HIP, std::string("RCCL ") + what + ": " + rccl().GetErrorString(r));
}

}  // namespace

#ifdef BFH_TEST_TRANSPORT
#include "comm_test_transport.hpp"   // (inside namespace bfh) libbuffalo_hip_test.so only
#else
// the product build: no test transport.  The environment knob is refused loudly instead of being ignored.
struct Comm::Shm {};
namespace {
bool shm_requested() {
    const char* t = std::getenv("BFH_COMM_TRANSPORT");
    if (t && std::string(t) == "shm")
        throw Error(BFH_ERR_UNSUPPORTED, "BFH_COMM_TRANSPORT=shm: the shared-memory TEST transport is not part of libbuffalo_hip.so "
                                         "(csrc/comm_test_transport.hpp, built into libbuffalo_hip_test.so with -DBFH_TEST_TRANSPORT)");
    return false;
}
}  // namespace
static void shm_unique_id(char*) {}
bool Comm::shm_attach(int, const char* id128) {
    BFH_REQUIRE(std::memcmp(id128, "BFHSHM1", 8) != 0, "comm_create: a shared-memory TEST transport id reached the product library (libbuffalo_hip_test.so has that transport)");
    return false;
}
template <typename T>
void Comm::shm_all_reduce(const T*, T*, size_t, hipStream_t) { throw Error(BFH_ERR_UNSUPPORTED, "no test transport in this build"); }
void Comm::shm_broadcast(void*, size_t, int, hipStream_t) { throw Error(BFH_ERR_UNSUPPORTED, "no test transport in this build"); }
#endif

void Comm::unique_id(char* out128) {
    if (shm_requested()) { shm_unique_id(out128); return; }
    ncclUniqueId id;
    check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(out128, &id, sizeof(id));
}

Comm::Comm(int n_ranks, int rank, const char* id128, int dev) {
    BFH_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks && id128, "comm_create: bad rank / size / id");
    device = dev;
    rank_ = rank;
    size_ = n_ranks;
    BFH_HIP(hipSetDevice(dev));
    BFH_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (shm_attach(n_ranks, id128)) return;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    check(rccl().CommInitRank(&c, n_ranks, id, rank), "ncclCommInitRank");
    comm_ = c;
    int count = 0;
    check(rccl().CommCount(c, &count), "ncclCommCount");
    BFH_REQUIRE(count == n_ranks, "comm_create: RCCL reports " + std::to_string(count) + " ranks, expected " + std::to_string(n_ranks));
}

// what the communicator that exists reports NOW: ncclCommCount of the live RCCL communicator (not the number it was asked for)
int Comm::live_size() const {
    if (!comm_) return size_;
    int count = 0;
    check(rccl().CommCount(static_cast<ncclComm_t>(comm_), &count), "ncclCommCount");
    return count;
}
std::string Comm::transport() const {
    if (shm_) return "shm-test";
    int v = 0;
    check(rccl().GetVersion(&v), "ncclGetVersion");
    return "rccl " + std::to_string(v / 10000) + "." + std::to_string((v / 100) % 100) + "." + std::to_string(v % 100);
}

Comm::~Comm() {
    delete shm_;
    if (comm_) (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm_));
    if (stream) (void)hipStreamDestroy(stream);
}

void Comm::all_reduce_f32(const float* send, float* recv, size_t count, hipStream_t s) {
    if (shm_) { shm_all_reduce(send, recv, count, s); return; }
    if (count) check(rccl().AllReduce(send, recv, count, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm_), s), "ncclAllReduce(f32)");
}
void Comm::all_reduce_i32(const int* send, int* recv, size_t count, hipStream_t s) {
    if (shm_) { shm_all_reduce(send, recv, count, s); return; }
    if (count) check(rccl().AllReduce(send, recv, count, ncclInt32, ncclSum, static_cast<ncclComm_t>(comm_), s), "ncclAllReduce(i32)");
}
void Comm::all_reduce_f64(const double* send, double* recv, size_t count, hipStream_t s) {
    if (shm_) { shm_all_reduce(send, recv, count, s); return; }
    if (count) check(rccl().AllReduce(send, recv, count, ncclFloat64, ncclSum, static_cast<ncclComm_t>(comm_), s), "ncclAllReduce(f64)");
}
void Comm::broadcast_bytes(void* buf, size_t bytes, int root, hipStream_t s) {
    if (shm_) { shm_broadcast(buf, bytes, root, s); return; }
    if (bytes) check(rccl().Broadcast(buf, buf, bytes, ncclInt8, root, static_cast<ncclComm_t>(comm_), s), "ncclBroadcast");
}
void Comm::group_start() { if (shm_) return; check(rccl().GroupStart(), "ncclGroupStart"); }
void Comm::group_end() { if (shm_) return; check(rccl().GroupEnd(), "ncclGroupEnd"); }

}  // namespace bfh

using bfh::Comm;
using bfh::guarded;

extern "C" {

int bfh_comm_unique_id(char* out, size_t bytes) {
    try {
        if (!out || bytes < 128) throw bfh::Error(BFH_ERR_INVALID, "comm_unique_id: need a 128-byte buffer");
        Comm::unique_id(out);
        return BFH_OK;
    } catch (const bfh::Error& e) {
        bfh::g_create_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        bfh::g_create_error = e.what();
        return BFH_ERR_HIP;
    }
}

void* bfh_comm_create(int n_ranks, int rank, const char* unique_id, int device) {
    try {
        return new Comm(n_ranks, rank, unique_id, device);
    } catch (const std::exception& e) {
        bfh::g_create_error = e.what();
        return nullptr;
    }
}

void bfh_comm_destroy(void* c) { delete static_cast<Comm*>(c); }
int bfh_comm_rank(void* c) { return c ? static_cast<Comm*>(c)->rank() : BFH_ERR_INVALID; }
int bfh_comm_size(void* c) {
    if (!c) return BFH_ERR_INVALID;
    int n = BFH_ERR_HIP;
    (void)guarded(c, [&] { n = static_cast<Comm*>(c)->live_size(); return BFH_OK; });
    return n;
}
int bfh_comm_transport(void* c, char* out, size_t bytes) {
    return guarded(c, [&] {
        BFH_REQUIRE(out && bytes > 0, "comm_transport: bad arguments");
        const std::string t = static_cast<Comm*>(c)->transport();
        std::snprintf(out, bytes, "%s", t.c_str());
        return BFH_OK;
    });
}

// every rank contributes rank + 1 per element; all must read n (n + 1) / 2 back
int bfh_comm_self_test(void* c) {
    return guarded(c, [&] {
        Comm* cm = static_cast<Comm*>(c);
        const int n = 1024;
        bfh::DevBuf<float> buf;
        buf.resize(n);
        std::vector<float> host(n, static_cast<float>(cm->rank() + 1));
        BFH_HIP(hipMemcpyAsync(buf.get(), host.data(), n * sizeof(float), hipMemcpyHostToDevice, cm->comm_stream()));
        cm->all_reduce_f32(buf.get(), buf.get(), n, cm->comm_stream());
        BFH_HIP(hipMemcpyAsync(host.data(), buf.get(), n * sizeof(float), hipMemcpyDeviceToHost, cm->comm_stream()));
        BFH_HIP(hipStreamSynchronize(cm->comm_stream()));
        const float want = 0.5f * cm->size() * (cm->size() + 1);
        for (float v : host)
            if (v != want) throw bfh::Error(BFH_ERR_HIP, "comm self test: all-reduce returned " + std::to_string(v) + ", expected " + std::to_string(want));
        return BFH_OK;
    });
}

// sum all-reduce of a host double array (loss sums, counters) through a staging buffer
int bfh_comm_all_reduce_f64(void* c, double* values, int n) {
    return guarded(c, [&] {
        Comm* cm = static_cast<Comm*>(c);
        BFH_REQUIRE(values && n >= 0, "comm_all_reduce_f64: bad arguments");
        if (n == 0) return BFH_OK;
        bfh::DevBuf<double> buf;
        buf.resize(static_cast<size_t>(n));
        BFH_HIP(hipMemcpyAsync(buf.get(), values, n * sizeof(double), hipMemcpyHostToDevice, cm->comm_stream()));
        cm->all_reduce_f64(buf.get(), buf.get(), static_cast<size_t>(n), cm->comm_stream());
        BFH_HIP(hipMemcpyAsync(values, buf.get(), n * sizeof(double), hipMemcpyDeviceToHost, cm->comm_stream()));
        BFH_HIP(hipStreamSynchronize(cm->comm_stream()));
        return BFH_OK;
    });
}

}  // extern "C"
