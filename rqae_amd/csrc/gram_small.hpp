// gram_small.hpp -- the few device helpers the matrix-core arm's translation units share (gram.hip, recon3.hip); included inside
// each unit's anonymous namespace.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// compile-time loop: the body is instantiated once per index (hipcc refuses to unroll the level loop by pragma)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}


inline size_t al256(size_t x) { return (x + 255) & ~size_t(255); }
__host__ __device__ constexpr size_t pad1k(size_t v) { return (v + 1023) & ~size_t(1023); }


__device__ __forceinline__ const void *uniform_ptr(const void *p)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned long long u = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a) |
                                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32);
    return reinterpret_cast<const void *>(u);
}

// one 1-KiB LDS-DMA piece: lane i moves 16 bytes from g_uniform + 16 i to LDS byte address lds_addr + 16 i (M0 saved/restored)
__device__ __forceinline__ void dma_piece(const void *g_uniform, unsigned lds_addr, unsigned lane_byte_off)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_off), "s"(g_uniform), "s"(lds_addr)
                 : "memory");
}

// the same piece with the non-temporal hint: the x stream is read once and should not push the weight images out of L2 / MALL
__device__ __forceinline__ void dma_piece_nt(const void *g_uniform, unsigned lds_addr, unsigned lane_byte_off)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_off), "s"(g_uniform), "s"(lds_addr)
                 : "memory");
}


__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// An LDS read the compiler does not schedule (16 bytes per lane at addr + OFF) and the counted wait that hands its result on: LDS
// returns in order, so "at most NEWER reads outstanding" means everything issued before the NEWER newest has arrived.  The wait
// takes the registers as operands -- whatever consumes them stays behind it.  (Left to itself hipcc reads an operand pair, waits out
// the LDS latency and multiplies; sched_group_barrier pipelines do not survive other vector work in the loop.)
template <int OFF>
__device__ __forceinline__ void lds_read16(u32x4 &dst, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int NEWER>
__device__ __forceinline__ void lds_wait_pair(u32x4 &a, u32x4 &b)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(NEWER));
}
