"""dev helper: build tools/ablate/qstamps.so = libgnnb with wall-clock stamps (10 ns ticks) inside k_gather_update_q: where the gather
waves and the chain waves of the fused conv half-pass spend their time, summed over all waves into a device array that
gnnb_dev_qstamps() copies out (tools/qstamps_run.py prints it).  Timing build only; results are the shipped ones."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _flat import flat_source
src = flat_source()

def rep(old, new, count=1):
    global src
    assert src.count(old) == count, (src.count(old), old)
    src = src.replace(old, new)

# counters: [variant][16]; variant = 3 for 32-node tiles, else SRC
rep('struct FArgs {', '__device__ unsigned long long g_qs[4][16];\nstruct FArgs {')
# sparse gather: stamp after the table build
rep('''                                                     int Hs, int Ws, int lane, float* ssum = nullptr) {
  const int g = lane >> 4, i = lane & 15;
  float sacc = 0.0f;                                  // ssum: see gather_tile_sparse''',
    '''                                                     int Hs, int Ws, int lane, float* ssum = nullptr, long long* tmid = nullptr) {
  const int g = lane >> 4, i = lane & 15;
  float sacc = 0.0f;                                  // ssum: see gather_tile_sparse''')
rep('''  constexpr int CS = 4 * GATHER_CHS16;
  const int npad = (n + CS - 1) / CS * CS;''', '''  constexpr int CS = 4 * GATHER_CHS16;
  if (tmid) *tmid = wall_clock64();
  const int npad = (n + CS - 1) / CS * CS;''')
rep('''                                                      f32x4 (&acc)[4], float& ssum) {
  ssum = 0.0f;''', '''                                                      f32x4 (&acc)[4], float& ssum, long long* tmid = nullptr) {
  ssum = 0.0f;''')
rep('''                           a.sout ? &ssum : nullptr);
    } else if (interior) gather_tile16<true>(acc, cmt, lds_ko, lds_kvo, a.g.K2, rsrc, uy, ux, a.g.Hs, a.g.Ws, lane);''',
    '''                           a.sout ? &ssum : nullptr, tmid);
    } else if (interior) gather_tile16<true>(acc, cmt, lds_ko, lds_kvo, a.g.K2, rsrc, uy, ux, a.g.Hs, a.g.Ws, lane);''')
# kernel: begin stamp
rep('''  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool SPARSE = SRC == 1, EMBED = SRC == 2;
  float* qbase = lds + PackUpdL3::FLOATS + (POST ? 6144 : 0);''', '''  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool SPARSE = SRC == 1, EMBED = SRC == 2;
  constexpr int QV = LANES == 32 ? 3 : SRC;
  const long long t_begin = wall_clock64();
  float* qbase = lds + PackUpdL3::FLOATS + (POST ? 6144 : 0);''')
# chain wave stamps
rep('''      if (!ok) { if (lane == 0) atomicOr(a.u.status, 2); return; }
    }
    // claims the next tile, copies it out of its ring slot, releases the slot, runs the chain
    for (;;) {
      int T = 0;''', '''      if (!ok) { if (lane == 0) atomicOr(a.u.status, 2); return; }
    }
    const long long t_staged = wall_clock64();
    long long c_wait = 0, c_chain = 0, c_tiles = 0;
    auto chain_out = [&]() {
      if (lane == 0) {
        atomicAdd(&g_qs[QV][8], (unsigned long long)(t_staged - t_begin));
        atomicAdd(&g_qs[QV][9], (unsigned long long)c_wait);
        atomicAdd(&g_qs[QV][10], (unsigned long long)c_chain);
        atomicAdd(&g_qs[QV][11], (unsigned long long)(wall_clock64() - t_begin));
        atomicAdd(&g_qs[QV][12], 1ull);
        atomicAdd(&g_qs[QV][13], (unsigned long long)c_tiles);
      }
    };
    // claims the next tile, copies it out of its ring slot, releases the slot, runs the chain
    for (;;) {
      const long long tw0 = wall_clock64();
      int T = 0;''')
rep('''      if (nvalid < 0) { if (lane == 0) atomicOr(a.u.status, 2); return; }
      if (nvalid == 0) return;
      q_chain<POST>(a, lds, ring, nvalid, lane, [&]() {''', '''      if (nvalid < 0) { if (lane == 0) atomicOr(a.u.status, 2); return; }
      const long long tw1 = wall_clock64();
      c_wait += tw1 - tw0;
      if (nvalid == 0) { chain_out(); return; }
      c_tiles += 1;
      q_chain<POST>(a, lds, ring, nvalid, lane, [&]() {''')
rep('''      if (nvalid < 32) return;                       // the last, partly filled tile
    }
  }''', '''      c_chain += wall_clock64() - tw1;
      if (nvalid < 32) { chain_out(); return; }                       // the last, partly filled tile
    }
  }''')
# gather wave stamps
rep('''  fetch(wg);
  for (long r = wg; r < nrounds && !stuck; r += nwg) {
    if (!nx.in) break;''', '''  fetch(wg);
  const long long t_loop0 = wall_clock64();
  long long c_fetch = 0, c_tab = 0, c_kloop = 0, c_push = 0, c_free = 0, c_gtiles = 0;
  for (long r = wg; r < nrounds && !stuck; r += nwg) {
    const long long g0 = wall_clock64();
    if (!nx.in) break;''')
rep('''    const bool need = tc.valid && node_is_live(lb, ub);
    if (!__any(need)) continue;
    const Ratio rt = compute_ratio(lb, ub);
    float ssum = 0.0f;
    Frag X;
    f32x4 acc[4];
    if (LANES == 32) gather_compute_tile<false, SPARSE>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, el, lane, X, ssum);
    else gather_compute_tile16<EMBED, SPARSE>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, ew, eb, lane, acc, ssum);
''', '''    const bool need = tc.valid && node_is_live(lb, ub);
    if (!__any(need)) { c_fetch += wall_clock64() - g0; continue; }
    const Ratio rt = compute_ratio(lb, ub);
    float ssum = 0.0f;
    Frag X;
    f32x4 acc[4];
    const long long g1 = wall_clock64();
    long long gm = g1;
    if (LANES == 32) gather_compute_tile<false, SPARSE>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, el, lane, X, ssum);
    else gather_compute_tile16<EMBED, SPARSE>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, ew, eb, lane, acc, ssum, &gm);
    // (make the stamp wait for the accumulators)
    if (LANES == 16 ? acc[0][0] == 12345.678f : X.t[0][0] == 12345.678f) a.u.status[0] |= 4;
    const long long g2 = wall_clock64();
    c_fetch += g1 - g0; c_tab += gm - g1; c_kloop += g2 - gm; c_gtiles += 1;
''')
rep('''      if (!ok) { if (lane == 0) atomicOr(a.u.status, 2); stuck = true; break; }
      const bool mine = need && (pos >> 5) == T;''', '''      if (!ok) { if (lane == 0) atomicOr(a.u.status, 2); stuck = true; break; }
      const bool mine = need && (pos >> 5) == T;''')
rep('''      if (lane == 0) __hip_atomic_fetch_add(&q->filled[s], hi - lo, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (lane == 0) __hip_atomic_fetch_add(&q->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}''', '''      if (lane == 0) __hip_atomic_fetch_add(&q->filled[s], hi - lo, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    c_push += wall_clock64() - g2;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (lane == 0) __hip_atomic_fetch_add(&q->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (lane == 0) {
    const long long t_end = wall_clock64();
    atomicAdd(&g_qs[QV][0], (unsigned long long)(t_loop0 - t_begin));
    atomicAdd(&g_qs[QV][1], (unsigned long long)c_fetch);
    atomicAdd(&g_qs[QV][2], (unsigned long long)c_tab);
    atomicAdd(&g_qs[QV][3], (unsigned long long)c_kloop);
    atomicAdd(&g_qs[QV][4], (unsigned long long)c_push);
    atomicAdd(&g_qs[QV][5], (unsigned long long)(t_end - t_begin));
    atomicAdd(&g_qs[QV][6], 1ull);
    atomicAdd(&g_qs[QV][7], (unsigned long long)c_gtiles);
  }
}''')
src += '''
extern "C" int gnnb_dev_qstamps(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qs), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
  if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_qs), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
'''
os.makedirs('/root/repo/tools/ablate', exist_ok=True)
open('/tmp/gnnb_qstamps.hip', 'w').write(src)
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-pthread', '-o', '/root/repo/tools/ablate/qstamps.so', '/tmp/gnnb_qstamps.hip'])
print("built tools/ablate/qstamps.so")
