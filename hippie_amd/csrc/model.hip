// hp_model_*: a serialised lowered model (".hpm" file written by hippie_amd/export.py) loaded, bound to device arenas the
// LIBRARY allocates, and stepped through the reference's verbs (forward / backward / optimizer step) — the part of the C ABI a
// host WITHOUT Python uses.  hp_program_* (program.hip) stays the lower level: caller-owned arenas + op records.
//
// Reference interface this stands under: constructing hippieUnimodalCVAE / MultiModalCVAE (hippie/model.py:13-44,352-395) and
// the Lightning automatic-optimisation loop around training_step (model.py:95-116): zero_grad -> training_step -> backward ->
// [clip] -> optimizer.step().
#include "hp_common.h"

#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

constexpr int kHpmVersion = 1;

struct HpmHeader {
  char magic[8];
  int32_t version, abi;
  int32_t n_ops, n_segments, n_params, n_bufs, n_io, has_init;
  int64_t arena_bytes[HP_NUM_SPACES];
  int32_t config[16];      // kind(0 unimodal / 1 multimodal), z_dim, output_size, output_size2, class_hidden_dim, num_sources, num_classes, batch, with_class, n_active, 0...
};
static_assert(sizeof(HpmHeader) == 8 + 8 * 4 + 6 * 8 + 16 * 4, "HpmHeader layout");

struct HpmSegment { char name[32]; int32_t first, count; };
static_assert(sizeof(HpmSegment) == 40, "HpmSegment layout");
static_assert(sizeof(HpTensorInfo) == 112 + 8 + 16 + 4 + 16 + 4, "HpTensorInfo layout");

thread_local std::string g_merr;

}  // namespace

// hp_last_error() is defined in program.hip; model errors are routed through it
namespace hp { int set_error(const std::string& msg); }

// Data parallelism for a host without Python: RCCL (librccl.so, the ROCm build of NCCL) bound at run time with dlopen — the library
// itself has no link-time dependency on it, and a single-GPU host never loads it.
struct HpDp {
  void* lib = nullptr;
  int (*get_unique_id)(void*) = nullptr;
  // ncclCommInitRank takes the 128-byte id BY VALUE
  struct Id { char b[128]; };
  int (*comm_init_rank)(void**, int, Id, int) = nullptr;
  int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*comm_destroy)(void*) = nullptr;
  const char* (*error_string)(int) = nullptr;
  void* comm = nullptr;
  hipStream_t cstream = nullptr;          // the communicator's stream: collectives run beside the model stream's kernels
  hipEvent_t ev_half[2] = {nullptr, nullptr}, ev_done = nullptr;
  int world = 0, rank = 0;
  int64_t dec_off = 0, cemb_off = 0, n_active = 0;      // floats: the decoder-side gradient bucket is [dec_off, cemb_off)
};

struct HpModel {
  HpmHeader hdr;
  std::vector<HpOp> ops;
  std::vector<HpmSegment> segments;
  std::vector<HpTensorInfo> tensors[3];          // params, buffers, io slots
  std::vector<float> init_param, init_buf, init_m, init_v;
  void* arenas[HP_NUM_SPACES] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  HpProgram* prog = nullptr;
  std::map<std::string, int> graphs;             // segment name -> captured segment id
  int64_t batches_tracked = 0;                   // BatchNorm num_batches_tracked (all layers advance together)
  bool on_device = false;
  HpDp* dp = nullptr;                            // hp_model_allreduce_init
};

namespace {

int merr(const std::string& msg) { return hp::set_error(msg); }

// hp_pick_side_stream: one wave that holds its hardware queue for `ticks` of the 100 MHz wall clock, and a kernel that does nothing
__global__ void hp_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void hp_empty_kernel() {}

bool dp_bind(HpDp* d, std::string& why) {
  const char* env = getenv("HIPPIE_RCCL_LIB");
  const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    if (!n || !n[0]) continue;
    d->lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (d->lib) break;
  }
  if (!d->lib) { why = "cannot load librccl.so (set HIPPIE_RCCL_LIB to its path)"; return false; }
  d->get_unique_id = reinterpret_cast<int (*)(void*)>(dlsym(d->lib, "ncclGetUniqueId"));
  d->comm_init_rank = reinterpret_cast<int (*)(void**, int, HpDp::Id, int)>(dlsym(d->lib, "ncclCommInitRank"));
  d->all_reduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(d->lib, "ncclAllReduce"));
  d->comm_destroy = reinterpret_cast<int (*)(void*)>(dlsym(d->lib, "ncclCommDestroy"));
  d->error_string = reinterpret_cast<const char* (*)(int)>(dlsym(d->lib, "ncclGetErrorString"));
  if (!d->get_unique_id || !d->comm_init_rank || !d->all_reduce || !d->comm_destroy) { why = "librccl.so lacks an expected symbol"; return false; }
  return true;
}
std::string dp_err(const HpDp* d, const char* what, int rc) {
  return std::string(what) + ": " + (d && d->error_string ? d->error_string(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
}
void dp_free(HpDp* d) {
  if (!d) return;
  if (d->comm && d->comm_destroy) d->comm_destroy(d->comm);
  for (hipEvent_t e : {d->ev_half[0], d->ev_half[1], d->ev_done}) if (e) hipEventDestroy(e);
  if (d->cstream) hipStreamDestroy(d->cstream);
  // (the library stays loaded: RCCL keeps process-wide state that does not survive a dlclose)
  delete d;
}

bool read_exact(FILE* f, void* dst, size_t n) { return n == 0 || fread(dst, 1, n, f) == n; }

const HpmSegment* find_segment(const HpModel* m, const char* name) {
  for (const auto& s : m->segments)
    if (strncmp(s.name, name, sizeof s.name) == 0) return &s;
  return nullptr;
}

int dtype_size(int dt) { return dt == 0 ? 4 : 8; }

}  // namespace

extern "C" {

namespace {
int model_load(const char* path, int flags, HpModel** out);
int model_save(HpModel* m, const char* path, int with_optimizer);
}
// (std::bad_alloc from a vector sized by a damaged header must not cross the C boundary: the library never aborts)
int hp_model_load(const char* path, int flags, HpModel** out) {
  try { return model_load(path, flags, out); } catch (const std::exception& ex) { return merr(std::string("hp_model_load: ") + ex.what()); }
}
int hp_model_save(HpModel* m, const char* path, int with_optimizer) {
  try { return model_save(m, path, with_optimizer); } catch (const std::exception& ex) { return merr(std::string("hp_model_save: ") + ex.what()); }
}
}  // extern "C"
namespace {
int model_load(const char* path, int flags, HpModel** out) {
  if (!path || !out) return merr("hp_model_load: null argument");
  FILE* f = fopen(path, "rb");
  if (!f) return merr(std::string("hp_model_load: cannot open ") + path);
  HpModel* m = new HpModel();
  // whatever way this function is left (an error return, an exception from a vector sized by a damaged header): file closed, model freed
  struct Guard { FILE*& f; HpModel*& m; ~Guard() { if (f) fclose(f); if (m) hp_model_destroy(m); } } guard{f, m};
  auto bail = [&](const std::string& why) { return merr("hp_model_load: " + why); };
  if (!read_exact(f, &m->hdr, sizeof m->hdr)) return bail("short file (header)");
  const HpmHeader& h = m->hdr;
  if (memcmp(h.magic, "HPMODEL", 8) != 0) return bail("not an .hpm file (bad magic)");
  if (h.version != kHpmVersion) return bail("unsupported file version " + std::to_string(h.version));
  if (h.abi != HP_ABI_VERSION) return bail("lowered for ABI " + std::to_string(h.abi) + ", this library is ABI " + std::to_string(HP_ABI_VERSION));
  if (h.n_ops <= 0 || h.n_ops > (1 << 20) || h.n_segments < 0 || h.n_segments > 4096 || h.n_params < 0 || h.n_params > (1 << 20) ||
      h.n_bufs < 0 || h.n_bufs > (1 << 20) || h.n_io < 0 || h.n_io > 4096)
    return bail("implausible table sizes");
  for (int k = 0; k < HP_NUM_SPACES; ++k)
    if (h.arena_bytes[k] <= 0 || (h.arena_bytes[k] & 3) || h.arena_bytes[k] > (int64_t(1) << 40)) return bail("bad arena size");
  m->ops.resize(h.n_ops);
  m->segments.resize(h.n_segments);
  if (!read_exact(f, m->ops.data(), sizeof(HpOp) * h.n_ops)) return bail("short file (ops)");
  if (!read_exact(f, m->segments.data(), sizeof(HpmSegment) * h.n_segments)) return bail("short file (segments)");
  const int counts[3] = {h.n_params, h.n_bufs, h.n_io};
  for (int w = 0; w < 3; ++w) {
    m->tensors[w].resize(counts[w]);
    if (!read_exact(f, m->tensors[w].data(), sizeof(HpTensorInfo) * counts[w])) return bail("short file (tensor table)");
    for (auto& t : m->tensors[w]) {
      t.name[sizeof t.name - 1] = 0;
      if (t.space < 0 || t.space >= HP_NUM_SPACES || t.offset_bytes < 0 || t.numel < 0 || t.dtype < 0 || t.dtype > 2 ||
          t.offset_bytes > h.arena_bytes[t.space] || t.numel > (h.arena_bytes[t.space] - t.offset_bytes) / dtype_size(t.dtype))
        return bail(std::string("tensor '") + t.name + "' lies outside its arena");
    }
  }
  if (h.has_init & 1) {
    m->init_param.resize(h.arena_bytes[HP_SPACE_PARAM] / 4);
    if (!read_exact(f, m->init_param.data(), h.arena_bytes[HP_SPACE_PARAM])) return bail("short file (parameter values)");
  }
  if (h.has_init & 2) {
    m->init_buf.resize(h.arena_bytes[HP_SPACE_BUF] / 4);
    if (!read_exact(f, m->init_buf.data(), h.arena_bytes[HP_SPACE_BUF])) return bail("short file (buffer values)");
  }
  if (h.has_init & 4) {      // AdamW moments (hp_model_save): exp_avg, exp_avg_sq in the parameter arena's layout
    m->init_m.resize(h.arena_bytes[HP_SPACE_M] / 4);
    m->init_v.resize(h.arena_bytes[HP_SPACE_V] / 4);
    if (!read_exact(f, m->init_m.data(), h.arena_bytes[HP_SPACE_M]) || !read_exact(f, m->init_v.data(), h.arena_bytes[HP_SPACE_V]))
      return bail("short file (optimiser moments)");
  }
  m->batches_tracked = (int64_t)((uint64_t)(uint32_t)h.config[13] | ((uint64_t)(uint32_t)h.config[14] << 32));
  fclose(f);
  f = nullptr;
  for (auto& s : m->segments) {
    s.name[sizeof s.name - 1] = 0;
    if (s.first < 0 || s.count < 0 || (int64_t)s.first + s.count > h.n_ops) { return merr("hp_model_load: segment out of range"); }
  }
  if (flags & HP_MODEL_NO_DEVICE) {
    // host-only: validate the records against the declared arena sizes (no allocation, no GPU)
    void* fake[HP_NUM_SPACES];
    for (int k = 0; k < HP_NUM_SPACES; ++k) fake[k] = nullptr;
    if (hp_program_create(m->ops.data(), h.n_ops, fake, h.arena_bytes, &m->prog)) { return 1; }
    *out = m;
    m = nullptr;
    return 0;
  }
  for (int k = 0; k < HP_NUM_SPACES; ++k) {
    hipError_t e = hipMalloc(&m->arenas[k], h.arena_bytes[k]);
    if (e == hipSuccess) e = hipMemset(m->arenas[k], 0, h.arena_bytes[k]);
    if (e != hipSuccess) { return merr(std::string("hp_model_load: arena allocation: ") + hipGetErrorString(e)); }
  }
  m->on_device = true;
  hipError_t e = hipSuccess;
  if (!m->init_param.empty()) e = hipMemcpy(m->arenas[HP_SPACE_PARAM], m->init_param.data(), h.arena_bytes[HP_SPACE_PARAM], hipMemcpyHostToDevice);
  if (e == hipSuccess && !m->init_buf.empty()) e = hipMemcpy(m->arenas[HP_SPACE_BUF], m->init_buf.data(), h.arena_bytes[HP_SPACE_BUF], hipMemcpyHostToDevice);
  if (e == hipSuccess && !m->init_m.empty()) e = hipMemcpy(m->arenas[HP_SPACE_M], m->init_m.data(), h.arena_bytes[HP_SPACE_M], hipMemcpyHostToDevice);
  if (e == hipSuccess && !m->init_v.empty()) e = hipMemcpy(m->arenas[HP_SPACE_V], m->init_v.data(), h.arena_bytes[HP_SPACE_V], hipMemcpyHostToDevice);
  if (e != hipSuccess) { return merr(std::string("hp_model_load: upload: ") + hipGetErrorString(e)); }
  if (hp_program_create(m->ops.data(), h.n_ops, m->arenas, h.arena_bytes, &m->prog)) { return 1; }
  *out = m;
  m = nullptr;
  return 0;
}

int model_save(HpModel* m, const char* path, int with_optimizer) {
  if (!m || !path) return merr("hp_model_save: null argument");
  if (!m->on_device) return merr("hp_model_save: the model was loaded with HP_MODEL_NO_DEVICE");
  hipError_t e = hipDeviceSynchronize();
  const int which[4] = {HP_SPACE_PARAM, HP_SPACE_BUF, HP_SPACE_M, HP_SPACE_V};
  std::vector<char> host[4];
  for (int k = 0; k < (with_optimizer ? 4 : 2) && e == hipSuccess; ++k) {
    host[k].resize(m->hdr.arena_bytes[which[k]]);
    e = hipMemcpy(host[k].data(), m->arenas[which[k]], host[k].size(), hipMemcpyDeviceToHost);
  }
  if (e != hipSuccess) return merr(std::string("hp_model_save: ") + hipGetErrorString(e));
  HpmHeader h = m->hdr;
  h.has_init = 3 | (with_optimizer ? 4 : 0);
  h.config[13] = (int32_t)(uint32_t)(m->batches_tracked & 0xFFFFFFFFll);      // int64, as the reference's num_batches_tracked: low / high word
  h.config[14] = (int32_t)(uint32_t)((uint64_t)m->batches_tracked >> 32);
  // written beside the target and renamed over it: a failed or interrupted write never destroys the previous checkpoint
  const std::string tmp = std::string(path) + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return merr(std::string("hp_model_save: cannot open ") + tmp);
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  ok = ok && fwrite(m->ops.data(), sizeof(HpOp), m->ops.size(), f) == m->ops.size();
  ok = ok && (m->segments.empty() || fwrite(m->segments.data(), sizeof(HpmSegment), m->segments.size(), f) == m->segments.size());
  for (int w = 0; w < 3 && ok; ++w)
    ok = m->tensors[w].empty() || fwrite(m->tensors[w].data(), sizeof(HpTensorInfo), m->tensors[w].size(), f) == m->tensors[w].size();
  for (int k = 0; k < (with_optimizer ? 4 : 2) && ok; ++k) ok = fwrite(host[k].data(), 1, host[k].size(), f) == host[k].size();
  ok = (fclose(f) == 0) && ok;
  if (!ok) { remove(tmp.c_str()); return merr(std::string("hp_model_save: write failed: ") + tmp); }
  if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return merr(std::string("hp_model_save: cannot rename onto ") + path); }
  return 0;
}
}  // namespace
extern "C" {

int hp_model_destroy(HpModel* m) {
  if (!m) return 0;
  if (m->dp) { if (m->on_device) (void)hipDeviceSynchronize(); dp_free(m->dp); m->dp = nullptr; }
  if (m->prog) hp_program_destroy(m->prog);
  for (int k = 0; k < HP_NUM_SPACES; ++k)
    if (m->arenas[k]) hipFree(m->arenas[k]);
  delete m;
  return 0;
}

int hp_model_config(const HpModel* m, int32_t out[16]) {
  if (!m || !out) return merr("hp_model_config: null argument");
  memcpy(out, m->hdr.config, sizeof m->hdr.config);
  return 0;
}

int hp_model_tensor_count(const HpModel* m, int which) {
  if (!m || which < 0 || which > 2) return -1;
  return (int)m->tensors[which].size();
}

int hp_model_tensor_info(const HpModel* m, int which, int index, HpTensorInfo* out) {
  if (!m || !out || which < 0 || which > 2 || index < 0 || index >= (int)m->tensors[which].size()) return merr("hp_model_tensor_info: bad argument");
  *out = m->tensors[which][index];
  return 0;
}

int hp_model_find(const HpModel* m, const char* name, HpTensorInfo* out) {
  if (!m || !name || !out) return merr("hp_model_find: null argument");
  for (int w = 2; w >= 0; --w)      // io slots first: "x", "src", "eps", "scalars", ...
    for (const auto& t : m->tensors[w])
      if (strcmp(t.name, name) == 0) { *out = t; return 0; }
  return merr(std::string("hp_model_find: no tensor named '") + name + "'");
}

void* hp_model_arena(const HpModel* m, int space, int64_t* nbytes) {
  if (!m || space < 0 || space >= HP_NUM_SPACES) return nullptr;
  if (nbytes) *nbytes = m->hdr.arena_bytes[space];
  return m->arenas[space];
}

HpProgram* hp_model_program(HpModel* m) { return m ? m->prog : nullptr; }

int hp_model_segment(const HpModel* m, const char* name, int* first, int* count) {
  if (!m || !name) return merr("hp_model_segment: null argument");
  const HpmSegment* s = find_segment(m, name);
  if (!s) return merr(std::string("hp_model_segment: no segment named '") + name + "'");
  if (first) *first = s->first;
  if (count) *count = s->count;
  return 0;
}

int hp_model_run(HpModel* m, const char* segment, int use_graph, void* stream) {
  if (!m || !segment) return merr("hp_model_run: null argument");
  if (!m->on_device) return merr("hp_model_run: the model was loaded with HP_MODEL_NO_DEVICE");
  const HpmSegment* s = find_segment(m, segment);
  if (!s) return merr(std::string("hp_model_run: no segment named '") + segment + "'");
  if (s->count == 0) return 0;
  if (!use_graph) return hp_program_run(m->prog, s->first, s->count, stream);
  auto it = m->graphs.find(segment);
  if (it == m->graphs.end()) {
    int id = -1;
    if (hp_program_capture(m->prog, s->first, s->count, &id)) return 1;
    it = m->graphs.emplace(segment, id).first;
  }
  return hp_program_replay(m->prog, it->second, stream);
}

// ---- the reference's verbs -------------------------------------------------------------------------------------------
int hp_model_forward(HpModel* m, int training, int use_graph, void* stream) {
  const int rc = hp_model_run(m, training ? "fwd_train" : "fwd_eval", use_graph, stream);
  if (rc == 0 && training) m->batches_tracked += 1;
  return rc;
}
int hp_model_backward(HpModel* m, int use_graph, void* stream) { return hp_model_run(m, "bwd", use_graph, stream); }
int hp_model_optimizer_step(HpModel* m, int use_graph, void* stream) { return hp_model_run(m, "opt", use_graph, stream); }
int hp_model_train_step(HpModel* m, int use_graph, void* stream) {
  if (m && find_segment(m, "step")) {
    const int rc = hp_model_run(m, "step", use_graph, stream);
    if (rc == 0) m->batches_tracked += 1;
    return rc;
  }
  int rc = hp_model_forward(m, 1, use_graph, stream);
  if (rc == 0) rc = hp_model_backward(m, use_graph, stream);
  if (rc == 0) rc = hp_model_optimizer_step(m, use_graph, stream);
  return rc;
}
int hp_model_train_step_staged(HpModel* m, int use_graph, void* stream) {
  if (!m) return merr("hp_model_train_step_staged: null argument");
  if (!find_segment(m, "stage")) return merr("hp_model_train_step_staged: the model was exported without resident tables (hippie_amd.export --resident-units N)");
  if (m->hdr.config[11] > 1)      // a data-parallel replica that skipped the gradient all-reduce would silently diverge from the others
    return merr("hp_model_train_step_staged: the model was exported for " + std::to_string(m->hdr.config[11]) +
                " data-parallel ranks; use hp_model_allreduce_init + hp_model_train_step_dp (or stage / fwd_train / bwd, your all-reduce, opt)");
  int rc;
  if (find_segment(m, "step_staged")) rc = hp_model_run(m, "step_staged", use_graph, stream);
  else {
    rc = hp_model_run(m, "stage", use_graph, stream);
    if (rc == 0) rc = hp_model_run(m, "fwd_train", use_graph, stream);
    if (rc == 0) rc = hp_model_run(m, "bwd", use_graph, stream);
    if (rc == 0) rc = hp_model_run(m, "opt", use_graph, stream);
  }
  if (rc == 0) m->batches_tracked += 1;
  return rc;
}
int hp_dp_unique_id(void* out128) {
  if (!out128) return merr("hp_dp_unique_id: null argument");
  HpDp d;
  std::string why;
  if (!dp_bind(&d, why)) return merr("hp_dp_unique_id: " + why);
  const int rc = d.get_unique_id(out128);
  if (rc != 0) return merr(dp_err(&d, "hp_dp_unique_id: ncclGetUniqueId", rc));
  return 0;
}

int hp_model_allreduce_init(HpModel* m, const void* unique_id, int rank, int world) {
  if (!m || !unique_id) return merr("hp_model_allreduce_init: null argument");
  if (!m->on_device) return merr("hp_model_allreduce_init: the model was loaded with HP_MODEL_NO_DEVICE");
  if (world < 1 || rank < 0 || rank >= world) return merr("hp_model_allreduce_init: need 0 <= rank < world");
  if (m->dp) return merr("hp_model_allreduce_init: the model already has a communicator");
  if (find_segment(m, "stage") && (m->hdr.config[11] != world || m->hdr.config[12] != rank))
    return merr("hp_model_allreduce_init: the model's resident tables were exported for rank " + std::to_string(m->hdr.config[12]) + " of " +
                std::to_string(m->hdr.config[11]) + " (hippie_amd.export --dp-world / --dp-rank)");
  HpDp* d = new HpDp();
  std::string why;
  if (!dp_bind(d, why)) { dp_free(d); return merr("hp_model_allreduce_init: " + why); }
  d->world = world; d->rank = rank;
  d->n_active = m->hdr.config[9];
  // the decoder-side gradient bucket = every parameter from the first "decoder*" key (decoder_fc.0.weight / decoder_fc_mod1.0.weight) up
  // to the class-embedding table, which is last in the arena (hippie_amd/planner.py: declaration order)
  int64_t dec = -1, cemb = d->n_active;
  for (const auto& t : m->tensors[0]) {
    const int64_t off = t.offset_bytes / 4;
    if (strncmp(t.name, "decoder", 7) == 0 && (dec < 0 || off < dec)) dec = off;
    if (strcmp(t.name, "class_embedding.weight") == 0) cemb = off < d->n_active ? off : d->n_active;
  }
  d->dec_off = dec; d->cemb_off = cemb;
  HpDp::Id id;
  memcpy(id.b, unique_id, sizeof id.b);
  int rc = d->comm_init_rank(&d->comm, world, id, rank);
  if (rc != 0) { const std::string msg = dp_err(d, "hp_model_allreduce_init: ncclCommInitRank", rc); d->comm = nullptr; dp_free(d); return merr(msg); }
  // (the communicator's stream is chosen at the first hp_model_train_step_dp, against the stream the caller steps this model on)
  hipError_t e = hipSuccess;
  for (hipEvent_t* ev : {&d->ev_half[0], &d->ev_half[1], &d->ev_done})
    if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
  if (e != hipSuccess) { dp_free(d); return merr(std::string("hp_model_allreduce_init: ") + hipGetErrorString(e)); }
  m->dp = d;
  return 0;
}

int hp_model_allreduce_destroy(HpModel* m) {
  if (!m || !m->dp) return 0;
  (void)hipDeviceSynchronize();
  dp_free(m->dp);
  m->dp = nullptr;
  return 0;
}

int hp_model_train_step_dp(HpModel* m, int use_graph, void* stream) {
  if (!m) return merr("hp_model_train_step_dp: null argument");
  HpDp* d = m->dp;
  if (!d) return merr("hp_model_train_step_dp: call hp_model_allreduce_init first");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* grad = static_cast<float*>(m->arenas[HP_SPACE_GRAD]);
  if (!d->cstream) {
    // A side stream that shares its hardware queue with the model's stream would turn every "wait for the backward half" into a barrier
    // in front of the model's own kernels (measured: 8.3 instead of 4.3 ms per pair-step): probed, once.
    void* busy[1] = {stream};
    void* c = nullptr;
    if (hp_pick_side_stream(busy, stream ? 1 : 0, 12, &c, nullptr)) return 1;
    d->cstream = static_cast<hipStream_t>(c);
  }
  // mean over the ranks of grad[lo, hi), in place, on the communicator's stream, after everything queued on `s` so far
  auto reduce = [&](hipEvent_t ev, std::initializer_list<std::pair<int64_t, int64_t>> ranges) -> int {
    hipError_t e = hipEventRecord(ev, s);
    if (e == hipSuccess) e = hipStreamWaitEvent(d->cstream, ev, 0);
    if (e != hipSuccess) return merr(std::string("hp_model_train_step_dp: ") + hipGetErrorString(e));
    for (const auto& r : ranges) {
      if (r.second <= r.first) continue;
      const int rc = d->all_reduce(grad + r.first, grad + r.first, (size_t)(r.second - r.first), /*ncclFloat32*/ 7, /*ncclAvg*/ 4, d->comm, d->cstream);
      if (rc != 0) return merr(dp_err(d, "hp_model_train_step_dp: ncclAllReduce", rc));
    }
    return 0;
  };
  int rc;
  if (find_segment(m, "fwd_train_staged")) rc = hp_model_run(m, "fwd_train_staged", use_graph, stream);
  else {
    rc = find_segment(m, "stage") ? hp_model_run(m, "stage", use_graph, stream) : 0;
    if (rc == 0) rc = hp_model_run(m, "fwd_train", use_graph, stream);
  }
  if (rc != 0) return rc;
  if (find_segment(m, "bwd_dec") && find_segment(m, "bwd_enc") && d->dec_off > 0) {
    // two halves, two buckets: the decoder-side bucket travels while the encoder-side half runs on the model's own stream
    rc = hp_model_run(m, "bwd_dec", use_graph, stream);
    if (rc == 0) rc = reduce(d->ev_half[0], {{d->dec_off, d->cemb_off}});
    if (rc == 0) rc = hp_model_run(m, "bwd_enc", use_graph, stream);
    if (rc == 0) rc = reduce(d->ev_half[1], {{0, d->dec_off}, {d->cemb_off, d->n_active}});
  } else {
    rc = hp_model_run(m, "bwd", use_graph, stream);
    if (rc == 0) rc = reduce(d->ev_half[0], {{0, d->n_active}});
  }
  if (rc != 0) return rc;
  hipError_t e = hipEventRecord(d->ev_done, d->cstream);
  if (e == hipSuccess) e = hipStreamWaitEvent(s, d->ev_done, 0);
  if (e != hipSuccess) return merr(std::string("hp_model_train_step_dp: ") + hipGetErrorString(e));
  rc = hp_model_run(m, "opt", use_graph, stream);
  if (rc == 0) m->batches_tracked += 1;
  return rc;
}

int64_t hp_model_batches_tracked(const HpModel* m) { return m ? m->batches_tracked : -1; }

int hp_model_set_optimizer(HpModel* m, float lr, float weight_decay, int reset_state) {
  if (!m) return merr("hp_model_set_optimizer: null argument");
  if (!(lr >= 0.f) || !(weight_decay >= 0.f)) return merr("hp_model_set_optimizer: lr and weight_decay must be >= 0");
  if (m->on_device) {        // the executor and its captured graphs are about to be replaced: nothing of them may still be running
    const hipError_t e0 = hipDeviceSynchronize();
    if (e0 != hipSuccess) return merr(std::string("hp_model_set_optimizer: ") + hipGetErrorString(e0));
  }
  int patched = 0;
  for (auto& op : m->ops)
    if (op.op == HP_OP_ADAMW) { op.f[0] = lr; op.f[4] = weight_decay; ++patched; }
  if (!patched) return merr("hp_model_set_optimizer: the program holds no HP_OP_ADAMW record");
  // the constants sit inside the executor's copy of the records and inside every captured graph: both are rebuilt
  // (hp_program_create validates again; graphs are captured again on their next use)
  HpProgram* fresh = nullptr;
  int64_t fake_bytes[HP_NUM_SPACES];
  void* bases[HP_NUM_SPACES];
  for (int k = 0; k < HP_NUM_SPACES; ++k) { fake_bytes[k] = m->hdr.arena_bytes[k]; bases[k] = m->on_device ? m->arenas[k] : nullptr; }
  if (hp_program_create(m->ops.data(), (int)m->ops.size(), bases, fake_bytes, &fresh)) return 1;
  if (m->prog) hp_program_destroy(m->prog);
  m->prog = fresh;
  m->graphs.clear();
  if (reset_state && m->on_device) {
    // a new optim.AdamW(model.parameters()) as the reference's module constructor makes (hippie/model.py:93): zero moments, step 0
    hipError_t e = hipMemset(m->arenas[HP_SPACE_M], 0, m->hdr.arena_bytes[HP_SPACE_M]);
    if (e == hipSuccess) e = hipMemset(m->arenas[HP_SPACE_V], 0, m->hdr.arena_bytes[HP_SPACE_V]);
    HpTensorInfo t;
    if (e == hipSuccess && hp_model_find(m, "adam_step", &t) == 0)
      e = hipMemset(static_cast<char*>(m->arenas[t.space]) + t.offset_bytes, 0, t.numel * dtype_size(t.dtype));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return merr(std::string("hp_model_set_optimizer: ") + hipGetErrorString(e));
  }
  return 0;
}

int hp_model_write(HpModel* m, const char* name, const void* src, int64_t nbytes, int src_on_device, void* stream) {
  HpTensorInfo t;
  if (!m || !src) return merr("hp_model_write: null argument");
  if (!m->on_device) return merr("hp_model_write: the model was loaded with HP_MODEL_NO_DEVICE");
  if (hp_model_find(m, name, &t)) return 1;
  if (nbytes != t.numel * dtype_size(t.dtype)) return merr(std::string("hp_model_write: '") + name + "' holds " + std::to_string(t.numel * dtype_size(t.dtype)) + " bytes, got " + std::to_string(nbytes));
  hipError_t e = hipMemcpyAsync(static_cast<char*>(m->arenas[t.space]) + t.offset_bytes, src, nbytes,
                                src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return merr(std::string("hp_model_write: ") + hipGetErrorString(e));
  return 0;
}

int hp_model_read(HpModel* m, const char* name, void* dst, int64_t nbytes, int dst_on_device, void* stream) {
  HpTensorInfo t;
  if (!m || !dst) return merr("hp_model_read: null argument");
  if (!m->on_device) return merr("hp_model_read: the model was loaded with HP_MODEL_NO_DEVICE");
  if (hp_model_find(m, name, &t)) return 1;
  if (nbytes != t.numel * dtype_size(t.dtype)) return merr(std::string("hp_model_read: '") + name + "' holds " + std::to_string(t.numel * dtype_size(t.dtype)) + " bytes, asked for " + std::to_string(nbytes));
  hipError_t e = hipMemcpyAsync(dst, static_cast<const char*>(m->arenas[t.space]) + t.offset_bytes, nbytes,
                                dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess && !dst_on_device) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return merr(std::string("hp_model_read: ") + hipGetErrorString(e));
  return 0;
}

int hp_model_synchronize(HpModel* m, void* stream) {
  (void)m;
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return merr(std::string("hp_model_synchronize: ") + hipGetErrorString(e));
  return 0;
}

// ---- streams for a host that holds two models ------------------------------------------------------------------------
int hp_stream_create(void** out) {
  if (!out) return merr("hp_stream_create: null argument");
  hipStream_t s = nullptr;
  const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) return merr(std::string("hp_stream_create: ") + hipGetErrorString(e));
  *out = s;
  return 0;
}

int hp_stream_destroy(void* stream) {
  if (!stream) return 0;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) return merr(std::string("hp_stream_destroy: ") + hipGetErrorString(e));
  return 0;
}

// events: what a host needs to bound how far it queues ahead of the GPU (bench.py Pair.run: two steps ahead is 1 % faster than everything at once)
int hp_event_create(void** out) {
  if (!out) return merr("hp_event_create: null argument");
  hipEvent_t e = nullptr;
  const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  if (rc != hipSuccess) return merr(std::string("hp_event_create: ") + hipGetErrorString(rc));
  *out = e;
  return 0;
}
int hp_event_record(void* event, void* stream) {
  const hipError_t rc = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
  if (rc != hipSuccess) return merr(std::string("hp_event_record: ") + hipGetErrorString(rc));
  return 0;
}
int hp_event_synchronize(void* event) {
  const hipError_t rc = hipEventSynchronize((hipEvent_t)event);
  if (rc != hipSuccess) return merr(std::string("hp_event_synchronize: ") + hipGetErrorString(rc));
  return 0;
}
int hp_event_destroy(void* event) {
  if (!event) return 0;
  const hipError_t rc = hipEventDestroy((hipEvent_t)event);
  if (rc != hipSuccess) return merr(std::string("hp_event_destroy: ") + hipGetErrorString(rc));
  return 0;
}

int hp_pick_side_stream(void* const* busy, int n_busy, int candidates, void** out, float report[2]) {
  if (!out || n_busy < 0 || (n_busy > 0 && !busy)) return merr("hp_pick_side_stream: bad argument");
  if (candidates < 1 || candidates > 32) return merr("hp_pick_side_stream: candidates must be 1..32");
  constexpr long long kSpinTicks = 200000;          // 2 ms
  hipEvent_t t0 = nullptr, t1 = nullptr;
  hipError_t e = hipEventCreate(&t0);
  if (e == hipSuccess) e = hipEventCreate(&t1);
  hipStream_t chosen = nullptr, least_bad = nullptr;
  float worst_of_chosen = 0.f, least_bad_ms = 1e30f;
  int tried = 0;
  for (int k = 0; k < candidates && e == hipSuccess && !chosen; ++k) {
    hipStream_t c = nullptr;
    e = hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
    if (e != hipSuccess) break;
    ++tried;
    float worst = 0.f;
    for (int j = 0; j < n_busy && e == hipSuccess; ++j) {
      hipStream_t sb = static_cast<hipStream_t>(busy[j]);
      // a kernel that occupies c's hardware queue, then a timed do-nothing kernel on the busy stream: on the same queue it waits its turn
      hipLaunchKernelGGL(hp_spin_kernel, dim3(1), dim3(64), 0, c, kSpinTicks);
      e = hipEventRecord(t0, sb);
      hipLaunchKernelGGL(hp_empty_kernel, dim3(1), dim3(64), 0, sb);
      if (e == hipSuccess) e = hipEventRecord(t1, sb);
      if (e == hipSuccess) e = hipEventSynchronize(t1);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, t0, t1);
      if (e == hipSuccess) e = hipStreamSynchronize(c);
      worst = ms > worst ? ms : worst;
    }
    if (e == hipSuccess && worst < 0.5f) { chosen = c; worst_of_chosen = worst; break; }      // (alone: ~0.01 ms; behind the spin: ~2 ms)
    if (e == hipSuccess && worst < least_bad_ms) {
      if (least_bad) hipStreamDestroy(least_bad);
      least_bad = c; least_bad_ms = worst;
    } else {
      hipStreamDestroy(c);
    }
  }
  if (t0) hipEventDestroy(t0);
  if (t1) hipEventDestroy(t1);
  if (e != hipSuccess) {
    if (chosen) hipStreamDestroy(chosen);
    if (least_bad) hipStreamDestroy(least_bad);
    return merr(std::string("hp_pick_side_stream: ") + hipGetErrorString(e));
  }
  if (!chosen) { chosen = least_bad; worst_of_chosen = least_bad_ms; least_bad = nullptr; }
  if (least_bad) hipStreamDestroy(least_bad);
  *out = chosen;
  if (report) { report[0] = worst_of_chosen * 1e3f; report[1] = (float)tried; }
  return 0;
}

int hp_pick_concurrent_streams(HpModel* a, HpModel* b, int candidates, float accept, void** stream_a, void** stream_b, float report[3]) {
  if (!a || !b || !stream_a || !stream_b) return merr("hp_pick_concurrent_streams: null argument");
  if (!a->on_device || !b->on_device) return merr("hp_pick_concurrent_streams: a model was loaded with HP_MODEL_NO_DEVICE");
  if (!find_segment(a, "fwd_eval") || !find_segment(b, "fwd_eval")) return merr("hp_pick_concurrent_streams: both models need a 'fwd_eval' segment");
  if (candidates < 2 || candidates > 16) return merr("hp_pick_concurrent_streams: candidates must be 2..16");
  if (!(accept > 0.f)) accept = 0.85f;
  constexpr int kReplays = 3;
  // pool[0]: normal priority, pool[1]: high priority (for the model whose graph takes longer alone: free-running, the shorter chain
  // otherwise finishes its steps first and the longer one runs its last steps alone — tools/micro/balance_probe.py)
  std::vector<hipStream_t> pool[2] = {std::vector<hipStream_t>(candidates, nullptr), std::vector<hipStream_t>(candidates, nullptr)};
  hipStream_t T = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};      // start, end, done a, done b
  int least = 0, greatest = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&T, hipStreamNonBlocking);
  for (int k = 0; k < candidates && e == hipSuccess; ++k) e = hipStreamCreateWithFlags(&pool[0][k], hipStreamNonBlocking);
  for (int k = 0; k < candidates && e == hipSuccess; ++k) e = hipStreamCreateWithPriority(&pool[1][k], hipStreamNonBlocking, greatest);
  for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreate(&ev[k]);
  int rc = e == hipSuccess ? 0 : merr(std::string("hp_pick_concurrent_streams: ") + hipGetErrorString(e));
  // the evaluation-forward graphs `replays` times, a on sa and / or b on sb (null: that model sits out), bracketed on the timing stream T
  auto timed = [&](hipStream_t sa, hipStream_t sb, int replays, float* us) -> int {
    hipError_t q = hipEventRecord(ev[0], T);
    if (q == hipSuccess && sa) q = hipStreamWaitEvent(sa, ev[0], 0);
    if (q == hipSuccess && sb && sb != sa) q = hipStreamWaitEvent(sb, ev[0], 0);
    if (q != hipSuccess) return merr(std::string("hp_pick_concurrent_streams: ") + hipGetErrorString(q));
    for (int r = 0; r < replays; ++r) {
      if (sa && hp_model_run(a, "fwd_eval", 1, sa)) return 1;
      if (sb && hp_model_run(b, "fwd_eval", 1, sb)) return 1;
    }
    if (sa) { q = hipEventRecord(ev[2], sa); if (q == hipSuccess) q = hipStreamWaitEvent(T, ev[2], 0); }
    if (q == hipSuccess && sb) { q = hipEventRecord(ev[3], sb); if (q == hipSuccess) q = hipStreamWaitEvent(T, ev[3], 0); }
    if (q == hipSuccess) q = hipEventRecord(ev[1], T);
    if (q == hipSuccess) q = hipEventSynchronize(ev[1]);
    float ms = 0.f;
    if (q == hipSuccess) q = hipEventElapsedTime(&ms, ev[0], ev[1]);
    if (q != hipSuccess) return merr(std::string("hp_pick_concurrent_streams: ") + hipGetErrorString(q));
    *us = ms * 1e3f / replays;
    return 0;
  };
  int bi = 0, bj = 0, tried = 0, pa = 0, pb = 0;
  float best = 1e30f, serial = 0.f;
  if (rc == 0) {
    float t = 0.f, alone[2] = {1e30f, 1e30f};
    rc = timed(pool[0][0], pool[0][0], 1, &t);                 // first replay of a graph uploads it
    for (int k = 0; k < 2 && rc == 0; ++k) {                   // a stream can be slow by itself: best of two
      rc = timed(pool[0][k], nullptr, kReplays, &t);
      if (rc == 0 && t < alone[0]) alone[0] = t;
      if (rc == 0) rc = timed(nullptr, pool[0][k], kReplays, &t);
      if (rc == 0 && t < alone[1]) alone[1] = t;
    }
    serial = alone[0] + alone[1];
    pa = alone[0] > alone[1] ? 1 : 0;                          // which pool each model draws from
    pb = 1 - pa;
    bool done = false;
    for (int j = 0; j < candidates && rc == 0 && !done; ++j)
      for (int i = 0; i < candidates && rc == 0 && !done; ++i) {
        rc = timed(pool[pa][i], pool[pb][j], kReplays, &t);
        ++tried;
        if (rc == 0 && t < best) { best = t; bi = i; bj = j; }
        done = rc == 0 && t <= accept * serial;
      }
  }
  for (int k = 0; k < 4; ++k) if (ev[k]) hipEventDestroy(ev[k]);
  if (T) hipStreamDestroy(T);
  for (int q = 0; q < 2; ++q)
    for (int k = 0; k < candidates; ++k) {
      const bool keep = rc == 0 && ((q == pa && k == bi) || (q == pb && k == bj));
      if (pool[q][k] && !keep) hipStreamDestroy(pool[q][k]);
    }
  if (rc != 0) return rc;
  *stream_a = pool[pa][bi];
  *stream_b = pool[pb][bj];
  if (report) { report[0] = best; report[1] = serial; report[2] = (float)tried; }
  return 0;
}

}  // extern "C"
