// The shared-memory TEST transport of Comm (comm.hpp), compiled ONLY with -DBFH_TEST_TRANSPORT: buffalo_amd/libbuffalo_hip_test.so, the library
// the N-ranks-on-one-GPU tests load (tests/comm_ranks_worker.py, bench.py with BFH_COMM_TRANSPORT=shm).  The product library
// (libbuffalo_hip.so) does not contain it: there BFH_COMM_TRANSPORT=shm is refused with an error that names this file.
// Included by comm.hip inside namespace bfh.
#pragma once

namespace {
constexpr char kShmMagic[8] = {'B', 'F', 'H', 'S', 'H', 'M', '1', 0};
constexpr size_t kShmSlot = size_t(4) << 20;      // bytes a rank stages per round
constexpr size_t kShmHeader = 4096;
constexpr int kShmMaxRanks = 16;
constexpr double kShmTimeoutS = 120.0;

struct ShmHeader {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> attached;
};

bool shm_requested() {
    const char* t = std::getenv("BFH_COMM_TRANSPORT");
    return t && std::string(t) == "shm";
}

}  // namespace

struct Comm::Shm {
    std::string name;
    int fd = -1;
    size_t bytes = 0;
    char* base = nullptr;
    ShmHeader* hdr = nullptr;
    std::vector<char> host;
    char* slot(int r) { return base + kShmHeader + static_cast<size_t>(r) * kShmSlot; }
    // central-counter barrier; the last rank to arrive opens the next generation
    void barrier(int n) {
        const uint32_t gen = hdr->generation.load(std::memory_order_acquire);
        if (hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == static_cast<uint32_t>(n)) {
            hdr->arrived.store(0, std::memory_order_relaxed);
            hdr->generation.fetch_add(1, std::memory_order_release);
            return;
        }
        const auto t0 = std::chrono::steady_clock::now();
        int spins = 0;
        while (hdr->generation.load(std::memory_order_acquire) == gen) {
            if (++spins < 2000) continue;
            std::this_thread::yield();
            if ((spins & 0xfff) == 0 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kShmTimeoutS)
                throw Error(BFH_ERR_HIP, "shm transport: a rank did not reach the collective within " + std::to_string(int(kShmTimeoutS)) +
                                             " s (ranks must issue the same collectives in the same order)");
        }
    }
    ~Shm() {
        bool last = false;
        if (hdr) last = hdr->attached.fetch_sub(1, std::memory_order_acq_rel) == 1;
        if (base) munmap(base, bytes);
        if (fd >= 0) close(fd);
        if (last) shm_unlink(name.c_str());
    }
};


// the id names the segment: magic + 16 random bytes
static void shm_unique_id(char* out128) {
        std::memset(out128, 0, 128);
        std::memcpy(out128, kShmMagic, sizeof(kShmMagic));
        int fd = open("/dev/urandom", O_RDONLY);
        if (fd < 0 || read(fd, out128 + 8, 16) != 16) {
            if (fd >= 0) close(fd);
            throw Error(BFH_ERR_HIP, "shm transport: /dev/urandom is not readable");
        }
        close(fd);
}

bool Comm::shm_attach(int n_ranks, const char* id128) {
    if (std::memcmp(id128, kShmMagic, sizeof(kShmMagic)) != 0) return false;
        BFH_REQUIRE(n_ranks <= kShmMaxRanks, "shm transport: at most 16 ranks");
        struct Guard {   // a throw below must not leak the mapping, the segment's name or the stream (the destructor of a half-built Comm never runs)
            Shm*& p; hipStream_t& st; bool armed = true, created = false;
            ~Guard() {
                if (!armed) return;
                if (p && created && !p->name.empty()) shm_unlink(p->name.c_str());   // never counted in `attached`: ~Shm would not unlink it
                delete p;
                p = nullptr;
                if (st) { (void)hipStreamDestroy(st); st = nullptr; }
            }
        } guard{shm_, stream};
        shm_ = new Shm();
        static const char* hex = "0123456789abcdef";
        shm_->name = "/bfh_";
        for (int i = 0; i < 16; ++i) {
            const unsigned char b = static_cast<unsigned char>(id128[8 + i]);
            shm_->name += hex[b >> 4];
            shm_->name += hex[b & 15];
        }
        shm_->bytes = kShmHeader + static_cast<size_t>(n_ranks) * kShmSlot;
        shm_->fd = shm_open(shm_->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (shm_->fd >= 0) guard.created = true;
        else shm_->fd = shm_open(shm_->name.c_str(), O_CREAT | O_RDWR, 0600);
        if (shm_->fd < 0 || ftruncate(shm_->fd, static_cast<off_t>(shm_->bytes)) != 0)   // a fresh segment reads as zeros: counters start at 0
            throw Error(BFH_ERR_HIP, "shm transport: cannot create " + shm_->name);
        void* m = mmap(nullptr, shm_->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, shm_->fd, 0);
        if (m == MAP_FAILED) throw Error(BFH_ERR_HIP, "shm transport: mmap failed");
        shm_->base = static_cast<char*>(m);
        shm_->hdr = reinterpret_cast<ShmHeader*>(m);
        shm_->hdr->attached.fetch_add(1, std::memory_order_acq_rel);
        guard.created = false;   // counted: from here ~Shm unlinks when the last rank leaves
        shm_->host.resize(kShmSlot);
        shm_->barrier(n_ranks);   // every rank is attached before the first collective
        guard.armed = false;
    return true;
}

// rounds of at most one slot per rank: D2H into the rank's slot | barrier | sum the slots | barrier | H2D.
// The sum runs in rank order, or -- BFH_COMM_SHM_ORDER=ring -- in the order a ring all-reduce produces: the round's elements are cut
// into N segments, and segment c is accumulated starting at rank c + 1 and ending at rank c (reduce-scatter), then handed to everybody
// (all-gather).  Different segments see different summation orders; every rank still receives the SAME bits, which is the property
// the exchange protocol relies on (tests/test_comm_ranks_gpu.py holds the replicas to bit-identity under both orders).
template <typename T>
void Comm::shm_all_reduce(const T* send, T* recv, size_t count, hipStream_t s) {
    BFH_HIP(hipStreamSynchronize(s));
    const size_t per = kShmSlot / sizeof(T);
    T* out = reinterpret_cast<T*>(shm_->host.data());
    static const bool ring = [] { const char* o = std::getenv("BFH_COMM_SHM_ORDER"); return o && std::string(o) == "ring"; }();
    for (size_t off = 0; off < count; off += per) {
        const size_t n = std::min(per, count - off);
        BFH_HIP(hipMemcpy(shm_->slot(rank_), send + off, n * sizeof(T), hipMemcpyDeviceToHost));
        shm_->barrier(size_);
        if (ring) {
            const size_t seg = (n + size_ - 1) / size_;
            for (int c = 0; c < size_; ++c) {
                const size_t lo = std::min(n, c * seg), hi = std::min(n, (c + 1) * seg);
                const T* a = reinterpret_cast<const T*>(shm_->slot((c + 1) % size_));
                for (size_t i = lo; i < hi; ++i) out[i] = a[i];
                for (int k = 2; k <= size_; ++k) {
                    const T* b = reinterpret_cast<const T*>(shm_->slot((c + k) % size_));
                    for (size_t i = lo; i < hi; ++i) out[i] += b[i];
                }
            }
        } else {
            const T* a = reinterpret_cast<const T*>(shm_->slot(0));
            for (size_t i = 0; i < n; ++i) out[i] = a[i];
            for (int r = 1; r < size_; ++r) {
                const T* b = reinterpret_cast<const T*>(shm_->slot(r));
                for (size_t i = 0; i < n; ++i) out[i] += b[i];
            }
        }
        shm_->barrier(size_);
        BFH_HIP(hipMemcpy(recv + off, out, n * sizeof(T), hipMemcpyHostToDevice));
    }
}


void Comm::shm_broadcast(void* buf, size_t bytes, int root, hipStream_t s) {
        BFH_REQUIRE(root >= 0 && root < size_, "broadcast: bad root");
        BFH_HIP(hipStreamSynchronize(s));
        char* p = static_cast<char*>(buf);
        for (size_t off = 0; off < bytes; off += kShmSlot) {
            const size_t n = std::min(kShmSlot, bytes - off);
            if (rank_ == root) BFH_HIP(hipMemcpy(shm_->slot(0), p + off, n, hipMemcpyDeviceToHost));
            shm_->barrier(size_);
            if (rank_ != root) BFH_HIP(hipMemcpy(p + off, shm_->slot(0), n, hipMemcpyHostToDevice));
            shm_->barrier(size_);
        }
}
