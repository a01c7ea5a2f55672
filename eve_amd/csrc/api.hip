// ABI version and thread-local error reporting for libeve_hip.so.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace eve {
static thread_local char g_err[512] = "";
thread_local const char* g_last_kernel = "";
int set_error(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: HIP error %d (%s)", where, (int)e, hipGetErrorString(e));
    return 100 + (int)e;
}
int set_error_msg(const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return 1;
}
}  // namespace eve

extern "C" int eve_abi_version(void) { return EVE_ABI_VERSION; }

namespace eve {
static eve_dispatch_config default_dispatch_config() {
    eve_dispatch_config c;
    memset(&c, 0, sizeof(c));
    c.struct_bytes = (int)sizeof(c);
    c.conv_impl_v1 = 0; c.conv_tile_big = 0; c.conv_halo = 1; c.conv_ws64 = 1; c.conv_wg8 = 1;
    c.conv_wg8_min_tiles = 224; c.conv_wg8_s2_min_tiles = 48; c.halo_persist = 1;
    c.wgrad_target_wgs = 0; c.wgrad_min_rows = 1536; c.wgrad_halo = 1; c.wgrad_wg8 = 1;
    c.wg64_th = 0; c.wg64_nreg = 0; c.wg64_fixed = 1;
    c.in_split = 1; c.in_min_threads = 512; c.in_stats_one_pass = 1; c.stem_split = 1; c.in_trunk_kernels = 1; c.stem_fused_wgrad = 1; c.stem_fwd_pairs = 1; c.in_big_planes = 1; c.conv1x1_stream = 1; c.conv3x3_stream = 1;
    c.wgrad_halo_min_m = 1ll << 20;
    c.cgru_seq_max_b = 384;
    c.cgru_scan = 1; c.small_linear = 1; c.tail_loss_node = 1; c.bucket_elems = 4 * 1024 * 1024; c.gate_wait_polls = 1 << 21;
    return c;
}
static void env_int(const char* name, int& field) { if (const char* e = getenv(name)) field = atoi(e); }
// the ONLY place the library reads the environment: once, when it is loaded
static eve_dispatch_config load_dispatch_config() {
    eve_dispatch_config c = default_dispatch_config();
    if (const char* e = getenv("EVE_CONV_IMPL")) c.conv_impl_v1 = (e[0] == 'v' && e[1] == '1') ? 1 : 0;
    if (const char* e = getenv("EVE_CONV_TILE")) c.conv_tile_big = e[0] == '2' ? 1 : 0;
    env_int("EVE_CONV_HALO", c.conv_halo); env_int("EVE_CONV_WS64", c.conv_ws64); env_int("EVE_CONV_WG8", c.conv_wg8);
    env_int("EVE_CONV_WG8_MIN_TILES", c.conv_wg8_min_tiles); env_int("EVE_CONV_WG8_S2_MIN_TILES", c.conv_wg8_s2_min_tiles);
    env_int("EVE_HALO_PERSIST", c.halo_persist); env_int("EVE_WGRAD_TARGET_WGS", c.wgrad_target_wgs);
    env_int("EVE_WGRAD_MIN_ROWS", c.wgrad_min_rows); env_int("EVE_WGRAD_HALO", c.wgrad_halo); env_int("EVE_WGRAD_WG8", c.wgrad_wg8);
    env_int("EVE_WG64_TH", c.wg64_th); env_int("EVE_WG64_NREG", c.wg64_nreg); env_int("EVE_WG64_FIXED", c.wg64_fixed);
    env_int("EVE_IN_SPLIT", c.in_split); env_int("EVE_IN_MIN_THREADS", c.in_min_threads);
    env_int("EVE_IN_STATS_ONE_PASS", c.in_stats_one_pass); env_int("EVE_STEM_SPLIT", c.stem_split);
    env_int("EVE_IN_TRUNK", c.in_trunk_kernels);
    env_int("EVE_STEM_FUSED_WGRAD", c.stem_fused_wgrad);
    env_int("EVE_STEM_FWD_PAIRS", c.stem_fwd_pairs);
    env_int("EVE_IN_BIG_PLANES", c.in_big_planes);
    env_int("EVE_CONV1X1_STREAM", c.conv1x1_stream);
    env_int("EVE_CONV3X3_STREAM", c.conv3x3_stream);
    env_int("EVE_CGRU_SEQ_MAX_B", c.cgru_seq_max_b);
    env_int("EVE_CGRU_SCAN", c.cgru_scan); env_int("EVE_SMALL_LINEAR", c.small_linear); env_int("EVE_TAIL_LOSS_NODE", c.tail_loss_node);
    env_int("EVE_BUCKET_ELEMS", c.bucket_elems); env_int("EVE_GATE_WAIT_POLLS", c.gate_wait_polls);
    if (const char* e = getenv("EVE_WGRAD_HALO_MIN_M")) c.wgrad_halo_min_m = atoll(e);
    if (c.wgrad_min_rows < 64) c.wgrad_min_rows = 64;
    if (c.in_min_threads < 64) c.in_min_threads = 64;
    if (c.bucket_elems < 1) c.bucket_elems = 1;
    if (c.gate_wait_polls < 1) c.gate_wait_polls = 1;
    return c;
}
eve_dispatch_config g_cfg = load_dispatch_config();
}  // namespace eve

extern "C" int eve_get_dispatch_config(eve_dispatch_config* out) {
    if (!out) return eve::set_error_msg("get_dispatch_config: null pointer");
    *out = eve::g_cfg;
    return 0;
}
extern "C" int eve_get_default_dispatch_config(eve_dispatch_config* out) {
    if (!out) return eve::set_error_msg("get_default_dispatch_config: null pointer");
    *out = eve::default_dispatch_config();
    return 0;
}
extern "C" int eve_set_dispatch_config(const eve_dispatch_config* cfg) {
    if (!cfg || cfg->struct_bytes != (int)sizeof(eve_dispatch_config)) return eve::set_error_msg("set_dispatch_config: struct size mismatch (ABI)");
    if (cfg->wgrad_min_rows < 64 || cfg->in_min_threads < 64) return eve::set_error_msg("set_dispatch_config: wgrad_min_rows / in_min_threads must be >= 64");
    if (cfg->bucket_elems < 1 || cfg->gate_wait_polls < 1) return eve::set_error_msg("set_dispatch_config: bucket_elems / gate_wait_polls must be >= 1");
    eve::g_cfg = *cfg;
    return 0;
}
extern "C" const char* eve_last_error(void) { return eve::g_err; }
extern "C" const char* eve_last_kernel(void) { return eve::g_last_kernel; }
