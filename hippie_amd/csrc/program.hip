// Program executor and C ABI of libhippie_hip.so (see include/hippie_hip.h).
#include "hp_common.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct HpProgram {
  std::vector<HpOp> ops;
  void* bases[HP_NUM_SPACES];
  int64_t sizes[HP_NUM_SPACES];
  std::vector<hipGraphExec_t> segs;
  std::vector<hipGraph_t> graphs;
  hipStream_t capture_stream = nullptr;
  struct Group { void* probs = nullptr; void* blocks = nullptr; int nblocks = 0; int ntaps = 0; int bf16 = 0; void* leaves = nullptr; int n_leaves = 0; };
  std::vector<Group> groups;          // indexed by op index (empty entries for ops without device tables)
  bool groups_ready = false;
};

namespace {
thread_local std::string g_err;

int fail(const std::string& msg) {
  g_err = msg;
  return 1;
}
}  // namespace
namespace hp { int set_error(const std::string& msg) { return fail(msg); } }      // (model.hip reports through hp_last_error too)
namespace {
int fail_hip(const char* what, hipError_t e) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return 1;
}

void free_groups(std::vector<HpProgram::Group>& groups) {
  for (auto& g : groups) {
    if (g.probs) hipFree(g.probs);
    if (g.blocks) hipFree(g.blocks);
    if (g.leaves) hipFree(g.leaves);
  }
  groups.clear();
}

// device tables of every WGRAD_GROUP op and small-leaf group (lazily: program creation / validation must work without a
// GPU).  Built into a temporary and swapped in only when complete: a failure half way frees what was built so far.
int ensure_groups(HpProgram* p) {
  if (p->groups_ready) return 0;
  std::vector<HpProgram::Group> groups(p->ops.size());
  for (size_t k = 0; k < p->ops.size(); ++k) {
    const HpOp& op = p->ops[k];
    const int ngroup = (op.flags >> HP_FLAG_GROUP_SHIFT) & HP_FLAG_GROUP_MASK;
    HpProgram::Group& g = groups[k];
    hipError_t e = hipSuccess;
    if (ngroup > 0) {
      g.n_leaves = ngroup + 1;
      e = hp::build_small_group(&p->ops[k - ngroup], ngroup + 1, p->bases, &g.leaves);
    } else if (op.op == HP_OP_WGRAD_GROUP) {
      g.ntaps = op.i[2];
      g.bf16 = (p->ops[op.i[0]].flags & HP_CONV_BF16) ? ((p->ops[op.i[0]].flags & HP_FLAG_ACT_BF16) ? 2 : 1) : (p->ops[op.i[0]].flags & HP_CONV_BF16X3) ? 3 : 0;
      e = hp::build_wgrad_group(&p->ops[op.i[0]], op.i[1], p->bases, &g.probs, &g.blocks, &g.nblocks);
    }
    if (e != hipSuccess) {
      free_groups(groups);
      return fail_hip("building group tables", e);
    }
  }
  p->groups.swap(groups);
  p->groups_ready = true;
  return 0;
}

// A run / capture / profile range must not cut through a launch unit: members execute at their closing record (PAIR,
// WGRAD_GROUP, small-leaf group), so a range holding members without their closing record would silently skip them and a
// range holding a closing record without its members would run ops outside the range.
int check_range(const HpProgram* p, int first, int count, const char* who) {
  const int last = first + count;       // exclusive
  for (int k = first; k < last; ++k) {
    const HpOp& op = p->ops[k];
    int lo = k, hi = k;                 // the records this closing record executes
    const int ngroup = (op.flags >> HP_FLAG_GROUP_SHIFT) & HP_FLAG_GROUP_MASK;
    if (ngroup > 0) lo = k - ngroup;
    else if (op.op == HP_OP_WGRAD_GROUP) { lo = op.i[0]; hi = op.i[0] + op.i[1] - 1; }
    else if (op.op == HP_OP_PAIR) { lo = op.i[0] < op.i[1] ? op.i[0] : op.i[1]; hi = op.i[0] < op.i[1] ? op.i[1] : op.i[0]; }
    else if (op.op == HP_OP_HEADS) { lo = op.i[0]; hi = op.i[0] + op.i[1] - 1; }
    if (lo < first || hi >= last)
      return fail(std::string(who) + ": the range cuts through the launch unit closed by op " + std::to_string(k));
  }
  // every member inside the range needs its closing record inside it too
  std::vector<char> covered(count > 0 ? count : 0, 0);
  for (int k = first; k < last; ++k) {
    const HpOp& op = p->ops[k];
    const int ngroup = (op.flags >> HP_FLAG_GROUP_SHIFT) & HP_FLAG_GROUP_MASK;
    if (ngroup > 0) for (int j = k - ngroup; j < k; ++j) covered[j - first] = 1;
    else if (op.op == HP_OP_WGRAD_GROUP) for (int j = op.i[0]; j < op.i[0] + op.i[1]; ++j) covered[j - first] = 1;
    else if (op.op == HP_OP_PAIR) { covered[op.i[0] - first] = 1; covered[op.i[1] - first] = 1; }
    else if (op.op == HP_OP_HEADS) for (int j = op.i[0]; j < op.i[0] + op.i[1]; ++j) covered[j - first] = 1;
  }
  for (int k = first; k < last; ++k)
    if ((p->ops[k].flags & HP_FLAG_MEMBER) && !covered[k - first])
      return fail(std::string(who) + ": member op " + std::to_string(k) + " is inside the range but its group / pair record is not");
  return 0;
}

hipError_t dispatch(const HpOp& op, void* const* bases, hipStream_t s) {
  switch (op.op) {
    case HP_OP_CONV_TAPS: return hp::launch_conv_taps(op, bases, s);
    case HP_OP_WGRAD_TAPS: return hp::launch_wgrad_taps(op, bases, s);
    default: return hp::launch_small(op, bases, s);
  }
}

hipError_t run_one(HpProgram* p, int k, hipStream_t s) {
  const HpOp& op = p->ops[k];
  if (op.flags & HP_FLAG_MEMBER) return hipSuccess;            // done by its group / pair launch
  if ((op.flags >> HP_FLAG_GROUP_SHIFT) & HP_FLAG_GROUP_MASK) {
    const HpProgram::Group& g = p->groups[k];
    return hp::launch_small_group(&p->ops[k - (g.n_leaves - 1)], g.leaves, g.n_leaves, s);
  }
  if (op.op == HP_OP_WGRAD_GROUP) {
    const HpProgram::Group& g = p->groups[k];
    return hp::launch_wgrad_group(g.ntaps, g.bf16, g.probs, g.blocks, g.nblocks, s);
  }
  if (op.op == HP_OP_PAIR) {
    const HpOp& a = p->ops[op.i[0]];
    const HpOp& b = p->ops[op.i[1]];
    return a.op == HP_OP_CONV_TAPS ? hp::launch_conv_pair(a, b, p->bases, s) : hp::launch_small_pair(a, b, p->bases, s);
  }
  if (op.op == HP_OP_HEADS) return hp::launch_heads(&p->ops[op.i[0]], op.i[1], op.i[2], p->bases, s);
  return dispatch(op, p->bases, s);
}

// Bytes each op touches behind every buffer slot it uses (0 = slot unused / unchecked): the planner fixes all
// offsets at lowering time, so a mis-lowered op is caught here, on the host, instead of writing out of bounds.
// Returns the number of (slot, bytes) pairs written.
int op_extents(const HpOp& op, int64_t (&need)[HP_OP_NB]) {
  for (auto& v : need) v = 0;
  const int32_t* I = op.i;
  const int64_t f4 = 4, f8 = 8;
  auto stat = [&](int64_t C) { return (int64_t)hp_stat_repl((int)C) * 2 * C * f8; };      // replicated slot of C channels
  auto rows_in = [&]() { return (int64_t)(I[3] > 0 ? I[0] / I[3] : 0) * I[4]; };      // (M / Lout) * Lin
  auto max_tapw = [&]() { int m = 0; for (int j = 0; j < I[9] && j < HP_MAX_TAPS; ++j) m = I[16 + j] > m ? I[16 + j] : m; return (int64_t)m + 1; };
  auto strided = [&](int64_t rows, int64_t ld, int64_t w) { return rows > 0 ? ((rows - 1) * ld + w) * f4 : 0; };
  switch (op.op) {
    case HP_OP_CONV_TAPS: {
      const int64_t M = I[0], N = I[1], K = I[2];
      const int64_t out_rows = I[28] > 0 ? (int64_t)(I[3] > 0 ? M / I[3] : 0) * I[28] : M;     // rows of the (taller) output tensor
      bool src0 = false, src1 = false;
      int64_t w0 = 0, w1 = 0;
      for (int j = 0; j < I[9] && j < HP_MAX_TAPS; ++j) {
        if (I[22 + j]) { src1 = true; w1 = I[16 + j] + 1 > w1 ? I[16 + j] + 1 : w1; }
        else { src0 = true; w0 = I[16 + j] + 1 > w0 ? I[16 + j] + 1 : w0; }
      }
      if (src0) { need[0] = rows_in() * K * f4; need[1] = w0 * N * K * f4; }
      if (src1) { need[10] = rows_in() * K * f4; need[11] = w1 * N * K * f4; }
      need[2] = out_rows * N * f4;
      if (op.flags & HP_CONV_IN_BN) { need[5] = need[6] = need[7] = need[8] = K * f4; need[12] = stat(K); need[13] = need[14] = 2 * K * f4; }
      if (op.flags & HP_CONV_EPI_BNRED) {
        const int64_t ON = out_rows * N * f4;
        need[17] = ON; need[18] = 2 * N * f4; need[20] = stat(N);
        if (op.buf[15] != HP_NULL) need[15] = ON;
        if (op.buf[16] != HP_NULL) need[16] = ON; else need[19] = 2 * N * f4;
        if (op.buf[21] != HP_NULL) { need[21] = ON; need[22] = 2 * N * f4; need[23] = stat(N); }
      }
      if (op.flags & 2) need[3] = N * f4;
      if (op.flags & 4) need[4] = stat(N);
      if (op.flags & 8) { need[5] = need[6] = need[7] = need[8] = N * f4; if (op.buf[9] != HP_NULL) need[9] = M * N * f4; }
      break;
    }
    case HP_OP_WGRAD_TAPS: {
      const int64_t M = I[0], N = I[1], K = I[2];
      need[0] = M * N * f4; need[1] = rows_in() * K * f4;
      if (op.flags & HP_CONV_IN_BN) need[3] = 2 * K * f4;
      need[2] = (op.flags & 1) ? max_tapw() * N * K * f4 : ((int64_t)(I[22] - 1) * I[24] + max_tapw() * N * K) * f4;
      break;
    }
    case HP_OP_SLAB_REDUCE: need[0] = ((int64_t)(I[1] - 1) * I[2] + I[0]) * f4; need[1] = (int64_t)I[0] * f4; break;
    case HP_OP_BN_APPLY: {
      const int64_t MC = (int64_t)I[0] * I[1] * f4, C = (int64_t)I[1] * f4;
      need[0] = need[1] = MC; if (I[3]) need[2] = stat(I[1]);
      need[3] = need[4] = need[5] = need[6] = C; need[7] = 2 * C;
      if (I[2]) need[8] = MC;
      if (I[2] == 2) { if (I[3]) need[9] = stat(I[1]); need[10] = need[11] = need[12] = need[13] = C; need[14] = 2 * C; }
      break;
    }
    case HP_OP_BN_BWD_REDUCE: {
      const int64_t MC = (int64_t)I[0] * I[1] * f4, C = (int64_t)I[1] * f4;
      need[0] = need[2] = need[3] = need[4] = MC; if (I[2]) need[1] = MC;
      need[5] = 2 * C; need[6] = stat(I[1]);
      if (I[3]) { need[7] = MC; need[8] = 2 * C; need[9] = stat(I[1]); }
      if (op.buf[2] == HP_NULL) { need[2] = 0; need[10] = 2 * C; }
      break;
    }
    case HP_OP_BN_BWD_APPLY: {
      const int64_t MC = (int64_t)I[0] * I[1] * f4, C = (int64_t)I[1] * f4;
      need[0] = need[1] = need[5] = MC; need[2] = 2 * C; need[3] = stat(I[1]); need[4] = need[6] = need[7] = C;
      break;
    }
    case HP_OP_STEM_FWD:
      need[0] = (int64_t)I[0] * I[1] * f4; need[1] = (int64_t)I[3] * 3 * f4; need[2] = (int64_t)I[0] * I[2] * I[3] * f4;
      if (op.buf[3] != HP_NULL) need[3] = stat(I[3]);
      break;
    case HP_OP_STEM_WGRAD:
      need[0] = (int64_t)I[0] * I[2] * I[3] * f4; need[1] = (int64_t)I[0] * I[1] * f4; need[2] = (int64_t)I[3] * 3 * f4; break;
    case HP_OP_POOL_FWD: case HP_OP_REPEAT_BWD: {
      const int big = op.op == HP_OP_POOL_FWD ? 0 : 0;
      need[big] = (int64_t)I[0] * I[1] * I[2] * f4;
      if (op.op == HP_OP_REPEAT_BWD) { if (I[3]) need[1] = need[0]; need[2] = (int64_t)I[0] * I[2] * f4; }
      else need[1] = (int64_t)I[0] * I[2] * f4;
      break;
    }
    case HP_OP_POOL_BWD: case HP_OP_REPEAT_FWD:
      need[0] = (int64_t)I[0] * I[2] * f4; need[1] = (int64_t)I[0] * I[1] * I[2] * f4; break;
    case HP_OP_CONCAT: {
      need[0] = (int64_t)I[0] * I[2] * f4;
      for (int j = 0; j < I[1] && j < 4; ++j) {
        const int kind = I[4 + 3 * j], w = I[5 + 3 * j], ld = I[6 + 3 * j];
        if (kind == 0) need[1 + 2 * j] = strided(I[0], ld, w);
        else if (kind == 1) { need[1 + 2 * j] = strided(I[16 + j], ld, w); need[2 + 2 * j] = (int64_t)I[0] * f8; }
      }
      break;
    }
    case HP_OP_EMB_BWD:
      need[0] = strided(I[0], I[2], I[3] + I[1]); need[1] = (int64_t)I[0] * f8; need[2] = (int64_t)I[4] * I[1] * f4; break;
    case HP_OP_LINEAR_FWD:
      need[0] = strided(I[0], I[3], I[2]); need[1] = (int64_t)I[1] * I[2] * f4; if (op.buf[2] != HP_NULL) need[2] = (int64_t)I[1] * f4;
      need[3] = strided(I[0], I[4], I[1]); if (I[6]) need[4] = stat(I[1]);
      break;
    case HP_OP_LINEAR_BWD_X:
      need[0] = strided(I[0], I[3], I[1]); need[1] = (int64_t)I[1] * I[2] * f4; need[2] = strided(I[0], I[4], I[2]);
      if (I[5]) need[3] = strided(I[0], I[6], I[2]);
      break;
    case HP_OP_LINEAR_BWD_W:
      need[0] = strided(I[0], I[3], I[1]); need[1] = strided(I[0], I[4], I[2]); need[2] = (int64_t)I[1] * I[2] * f4;
      if (op.buf[3] != HP_NULL) need[3] = (int64_t)I[1] * f4;
      break;
    case HP_OP_REPARAM_KL_FWD:
      need[0] = (int64_t)I[0] * 2 * I[1] * f4; need[1] = need[2] = (int64_t)I[0] * I[1] * f4; need[3] = 4 * f8; break;
    case HP_OP_REPARAM_KL_BWD:
      need[0] = need[3] = (int64_t)I[0] * 2 * I[1] * f4; need[1] = (int64_t)I[0] * I[1] * f4; need[2] = strided(I[0], I[2], I[1]); break;
    case HP_OP_MSE_FWD_BWD: need[0] = need[1] = need[2] = (int64_t)I[0] * f4; need[3] = 4 * f8; break;
    case HP_OP_TAIL_FWD:
      need[0] = (int64_t)I[0] * I[1] * I[2] * f4; need[1] = (int64_t)I[2] * 3 * f4; need[2] = f4; need[3] = (int64_t)I[0] * 2 * I[1] * f4; break;
    case HP_OP_TAIL_BWD_X:
      need[0] = (int64_t)I[0] * 2 * I[1] * f4; need[1] = (int64_t)I[2] * 3 * f4; need[2] = (int64_t)I[0] * I[1] * I[2] * f4; break;
    case HP_OP_TAIL_BWD_W:
      need[0] = (int64_t)I[0] * 2 * I[1] * f4; need[1] = (int64_t)I[0] * I[1] * I[2] * f4; need[2] = (int64_t)I[2] * 3 * f4; need[3] = f4; break;
    case HP_OP_LOSS_FINALIZE: need[0] = 4 * f8; need[1] = 4 * f4; break;
    case HP_OP_GRADNORM: need[0] = (int64_t)I[0] * f4; need[1] = f8; break;
    case HP_OP_ADAMW:
      need[0] = need[1] = need[2] = need[3] = (int64_t)I[0] * f4; need[4] = f8; if (op.f[5] > 0.f) need[5] = f8; break;
    case HP_OP_ADAMW_SF:
      need[0] = need[1] = need[2] = need[3] = (int64_t)I[0] * f4; need[4] = f8; need[5] = 4 * f8; if (op.f[4] > 0.f) need[6] = f8; break;
    case HP_OP_SF_SCHEDULE: need[0] = f8; need[1] = 4 * f8; break;
    case HP_OP_LERP: need[0] = need[1] = (int64_t)I[0] * f4; break;
    case HP_OP_STEP_INC: need[0] = f8; break;
    case HP_OP_ZERO: need[0] = (int64_t)(uint32_t)I[0] + ((int64_t)(uint32_t)I[1] << 32); break;
    case HP_OP_RESAMPLE_LINEAR: need[0] = (int64_t)I[0] * I[1] * f4; need[1] = (int64_t)I[0] * I[2] * f4; break;
    case HP_OP_STATS_SYNC: need[0] = (int64_t)I[0] * f8; break;
    case HP_OP_STAGE_BATCH: {
      const int64_t B = I[0], L = I[1], L2 = I[2], z = I[3], N = I[7];
      need[0] = N * L * f4; if (L2 > 0) need[1] = N * L2 * f4;
      need[2] = N * f8; need[3] = (int64_t)I[4] * I[5] * B * f8; need[4] = f8;
      need[5] = B * L * f4; if (L2 > 0) need[6] = B * L2 * f4;
      need[7] = B * f8; need[8] = B * z * f4; need[9] = f8;
      break;
    }
    case HP_OP_WFRAG: {
      const int64_t T = I[0], N = I[1], K = I[2];
      need[0] = T * N * K * f4;
      if (I[3] & 1) need[1] = T * (K / 16) * ((N + 31) / 32) * 3072;
      if (I[3] & 2) need[2] = T * (((N + 31) / 32) * 2) * (K / 32) * 3072;
      break;
    }
    default: break;
  }
  if (op.op == HP_OP_CONV_TAPS && (op.flags & HP_CONV_WFRAG)) {
    // fragment images of the weights this op multiplies with: [slabs][K/16][ceil(N/32)] chunks of 3072 bytes
    const int64_t chunks = (int64_t)(I[2] / 16) * ((I[1] + 31) / 32) * 3072;
    if (need[1] > 0) need[24] = need[1] / ((int64_t)I[1] * I[2] * f4) * chunks;
    if (need[11] > 0) need[25] = need[11] / ((int64_t)I[1] * I[2] * f4) * chunks;
  }
  if (op.flags & HP_FLAG_ACT_BF16) {
    // the op's activation-typed buffers hold bf16: half the bytes (which buffers: include/hippie_hip.h, HP_FLAG_ACT_BF16)
    static const struct { int op; int slots[8]; } kAct[] = {
      {HP_OP_CONV_TAPS, {0, 10, 2, 9, 15, 16, 17, 21}}, {HP_OP_WGRAD_TAPS, {0, 1, -1}}, {HP_OP_BN_APPLY, {0, 1, 8, -1}},
      {HP_OP_BN_BWD_REDUCE, {0, 1, 2, 3, 4, 7, -1}}, {HP_OP_BN_BWD_APPLY, {0, 1, 5, -1}}, {HP_OP_STEM_FWD, {2, -1}}, {HP_OP_STEM_WGRAD, {0, -1}},
      {HP_OP_POOL_FWD, {0, -1}}, {HP_OP_POOL_BWD, {1, -1}}, {HP_OP_REPEAT_FWD, {1, -1}}, {HP_OP_REPEAT_BWD, {0, 1, -1}},
      {HP_OP_TAIL_FWD, {0, -1}}, {HP_OP_TAIL_BWD_X, {2, -1}}, {HP_OP_TAIL_BWD_W, {1, -1}}};
    for (const auto& e : kAct)
      if (e.op == op.op)
        for (int k = 0; k < 8 && e.slots[k] >= 0; ++k) need[e.slots[k]] /= 2;
  }
  return 0;
}

int validate_op(const HpOp& op, const int64_t* sizes, int index, std::string& why) {
  char buf[256];
  if (op.op <= 0 || op.op >= HP_OP__COUNT) {
    snprintf(buf, sizeof buf, "op %d: unknown opcode %d", index, op.op);
    why = buf;
    return 1;
  }
  int64_t need[HP_OP_NB];
  op_extents(op, need);
  for (int k = 0; k < HP_OP_NB; ++k) {
    const int64_t r = op.buf[k];
    if (r == HP_NULL) {
      if (need[k] > 0) {
        snprintf(buf, sizeof buf, "op %d (opcode %d): buffer slot %d is required (%lld bytes) but NULL", index, op.op, k, (long long)need[k]);
        why = buf;
        return 1;
      }
      continue;
    }
    const int sp = hp::space_of(r);
    const int64_t off = hp::offset_of(r);
    if (sp < 0 || sp >= HP_NUM_SPACES || off < 0 || off >= sizes[sp] || (off & 3) || need[k] < 0 || off + need[k] > sizes[sp]) {
      snprintf(buf, sizeof buf, "op %d (opcode %d): buffer slot %d out of range (space %d offset %lld extent %lld arena size %lld)", index,
               op.op, k, sp, (long long)off, (long long)need[k], (long long)(sp >= 0 && sp < HP_NUM_SPACES ? sizes[sp] : -1));
      why = buf;
      return 1;
    }
  }
  if (op.op == HP_OP_CONV_TAPS || op.op == HP_OP_WGRAD_TAPS) {
    const int M = op.i[0], N = op.i[1], K = op.i[2], nt = op.i[9];
    bool ok = M > 0 && N > 0 && K > 0 && (K % 4) == 0 && (N % 4) == 0 && op.i[3] > 0 && op.i[4] > 0 && nt >= 1 &&
              nt <= HP_MAX_TAPS && op.i[6] >= 1 && op.i[7] >= 0 && op.i[7] <= 1;
    if (op.op == HP_OP_CONV_TAPS) {
      ok = ok && (K % 32) == 0 && op.i[28] >= 0;
      if (op.i[28] > 0) ok = ok && op.i[29] >= 1 && op.i[30] >= 0 && (int64_t)op.i[29] * (op.i[3] - 1) + op.i[30] < op.i[28];
      if (op.flags & HP_CONV_IN_BN) ok = ok && K <= 512 && op.i[31] > 0 && !(op.flags & HP_CONV_BN_EVAL) && op.f[2] >= 0.f && op.f[2] <= 1.f;     // coefficient table in LDS; leaky_relu evaluated as max(v, v * slope)
      if (op.flags & HP_CONV_EPI_BNRED) ok = ok && !(op.flags & (HP_CONV_BIAS | HP_CONV_STATS | HP_CONV_BN_EVAL));
      for (int j = 0; j < nt && j < HP_MAX_TAPS; ++j) ok = ok && (op.i[22 + j] == 0 || op.i[22 + j] == 1);
    }
    ok = ok && !((op.flags & HP_CONV_BF16) && (op.flags & HP_CONV_BF16X3));      // one matrix mode per op
    if (op.op == HP_OP_WGRAD_TAPS)
      ok = ok && (nt == 1 || nt == 3) && op.i[22] >= 1 && op.i[23] > 0 && (op.i[23] % 32) == 0 &&
           (int64_t)op.i[22] * op.i[23] >= M && (!(op.flags & HP_CONV_IN_BN) || (op.f[0] >= 0.f && op.f[0] <= 1.f));
    if (!ok) {
      snprintf(buf, sizeof buf, "op %d (opcode %d): bad tap-map shape M=%d N=%d K=%d ntaps=%d", index, op.op, M, N, K, nt);
      why = buf;
      return 1;
    }
  }
  if (op.flags & HP_FLAG_ACT_BF16) {
    const int o = op.op;
    const bool typed = o == HP_OP_CONV_TAPS || o == HP_OP_WGRAD_TAPS || o == HP_OP_BN_APPLY || o == HP_OP_BN_BWD_REDUCE || o == HP_OP_BN_BWD_APPLY ||
                       o == HP_OP_STEM_FWD || o == HP_OP_STEM_WGRAD || o == HP_OP_POOL_FWD || o == HP_OP_POOL_BWD || o == HP_OP_REPEAT_FWD ||
                       o == HP_OP_REPEAT_BWD || o == HP_OP_TAIL_FWD || o == HP_OP_TAIL_BWD_X || o == HP_OP_TAIL_BWD_W;
    const bool bn = o == HP_OP_BN_APPLY || o == HP_OP_BN_BWD_REDUCE || o == HP_OP_BN_BWD_APPLY;
    const bool mm = o == HP_OP_CONV_TAPS || o == HP_OP_WGRAD_TAPS;
    if (!typed || (bn && (op.i[1] % 4) != 0) || (mm && !(op.flags & HP_CONV_BF16)) || (mm && ((op.i[1] % 8) != 0 || (op.i[2] % 8) != 0))) {
      snprintf(buf, sizeof buf, "op %d (opcode %d): HP_FLAG_ACT_BF16 needs an op with activation-typed buffers (BatchNorm ops: channels %% 4 == 0; "
               "CONV_TAPS / WGRAD_TAPS: HP_CONV_BF16, N and K multiples of 8)", index, op.op);
      why = buf;
      return 1;
    }
  }
  if (op.op == HP_OP_WFRAG && (op.i[0] <= 0 || op.i[1] <= 0 || op.i[2] <= 0 || (op.i[1] % 4) != 0 || (op.i[2] % 32) != 0 || (op.i[3] & ~3) != 0 || (op.i[3] & 3) == 0 ||
                               ((op.i[3] & 2) && (op.i[1] % 32) != 0))) {
    snprintf(buf, sizeof buf, "op %d: bad WFRAG shape (T=%d N=%d K=%d which=%d: N %% 4, K %% 32, the G form N %% 32)", index, op.i[0], op.i[1], op.i[2], op.i[3]);
    why = buf;
    return 1;
  }
  if (op.op == HP_OP_CONV_TAPS && (op.flags & HP_CONV_WFRAG) && (!(op.flags & HP_CONV_BF16X3) || (op.flags & HP_FLAG_ACT_BF16))) {
    snprintf(buf, sizeof buf, "op %d: HP_CONV_WFRAG needs HP_CONV_BF16X3 (and fp32-stored activations)", index);
    why = buf;
    return 1;
  }
  if (op.op == HP_OP_STAGE_BATCH && (op.i[0] <= 0 || op.i[1] <= 0 || op.i[2] < 0 || op.i[3] <= 0 || op.i[4] <= 0 || op.i[5] <= 0 || op.i[6] < 0 ||
                                     op.i[6] >= op.i[5] || op.i[7] <= 0)) {
    snprintf(buf, sizeof buf, "op %d: bad STAGE_BATCH shape (B=%d L=%d L2=%d z=%d batches=%d world=%d rank=%d N=%d)", index, op.i[0], op.i[1], op.i[2],
             op.i[3], op.i[4], op.i[5], op.i[6], op.i[7]);
    why = buf;
    return 1;
  }
  if (op.op == HP_OP_WGRAD_GROUP && (op.i[0] < 0 || op.i[1] <= 0 || op.i[0] + op.i[1] > index || (op.i[2] != 1 && op.i[2] != 3))) {
    snprintf(buf, sizeof buf, "op %d: bad wgrad group range [%d, +%d) taps %d", index, op.i[0], op.i[1], op.i[2]);
    why = buf;
    return 1;
  }
  if ((op.op == HP_OP_BN_APPLY || op.op == HP_OP_BN_BWD_REDUCE || op.op == HP_OP_BN_BWD_APPLY) &&
      (op.i[0] <= 0 || op.i[1] <= 0)) {
    snprintf(buf, sizeof buf, "op %d (opcode %d): bad M/C", index, op.op);
    why = buf;
    return 1;
  }
  return 0;
}
}  // namespace

extern "C" {

int hp_abi_version(void) { return HP_ABI_VERSION; }
const char* hp_last_error(void) { return g_err.c_str(); }

int hp_device_info(int* n_cu, int* wave_size, char* arch, int arch_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail_hip("hipGetDevice", e);
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return fail_hip("hipGetDeviceProperties", e);
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (wave_size) *wave_size = prop.warpSize;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return 0;
}

int hp_program_create(const HpOp* ops, int n_ops, void* const bases[HP_NUM_SPACES], const int64_t sizes[HP_NUM_SPACES],
                      HpProgram** out) {
  if (!ops || n_ops <= 0 || !bases || !sizes || !out) return fail("hp_program_create: null argument");
  HpProgram* p = new HpProgram();
  p->ops.assign(ops, ops + n_ops);
  for (int k = 0; k < HP_NUM_SPACES; ++k) { p->bases[k] = bases[k]; p->sizes[k] = sizes[k]; }
  if (hp_program_validate(p) != 0) { delete p; return 1; }
  *out = p;
  return 0;
}

int hp_program_destroy(HpProgram* p) {
  if (!p) return 0;
  // replays of these graphs (and kernels reading the group tables) may still be in flight on the caller's streams.  Only a program that
  // has device state of its own waits: one that was only validated (null arenas: HP_MODEL_NO_DEVICE) or never captured / grouped anything
  // must neither initialise the HIP runtime nor stall other programs' streams.
  bool device = false;
  for (int k = 0; k < HP_NUM_SPACES; ++k) device = device || p->bases[k] != nullptr;
  if (device && (!p->segs.empty() || p->groups_ready || p->capture_stream)) (void)hipDeviceSynchronize();
  for (auto g : p->segs) if (g) hipGraphExecDestroy(g);
  for (auto g : p->graphs) if (g) hipGraphDestroy(g);
  free_groups(p->groups);
  if (p->capture_stream) hipStreamDestroy(p->capture_stream);
  delete p;
  return 0;
}

int hp_program_validate(const HpProgram* p) {
  if (!p) return fail("hp_program_validate: null program");
  std::string why;
  for (size_t k = 0; k < p->ops.size(); ++k) {
    if (validate_op(p->ops[k], p->sizes, (int)k, why)) return fail(why);
    if (p->ops[k].op == HP_OP_PAIR) {
      const HpOp& g = p->ops[k];
      const bool in_range = g.i[0] >= 0 && g.i[1] >= 0 && g.i[0] < (int)k && g.i[1] < (int)k && g.i[0] != g.i[1];
      if (!in_range) return fail("pair op " + std::to_string(k) + ": member index out of range");
      const HpOp& a = p->ops[g.i[0]];
      const HpOp& b = p->ops[g.i[1]];
      const bool kind_ok = a.op == b.op && (a.flags & HP_FLAG_MEMBER) && (b.flags & HP_FLAG_MEMBER) &&
                           (a.flags & HP_FLAG_ACT_BF16) == (b.flags & HP_FLAG_ACT_BF16) &&
                           ((a.op == HP_OP_CONV_TAPS && (a.flags & 1) == (b.flags & 1) && (a.flags & (HP_CONV_BF16 | HP_CONV_BF16X3 | HP_CONV_WFRAG)) == (b.flags & (HP_CONV_BF16 | HP_CONV_BF16X3 | HP_CONV_WFRAG))) ||
                            ((a.op == HP_OP_BN_APPLY || a.op == HP_OP_BN_BWD_REDUCE || a.op == HP_OP_BN_BWD_APPLY) &&
                             (a.i[1] % 4 == 0) == (b.i[1] % 4 == 0)));
      if (!kind_ok) return fail("pair op " + std::to_string(k) + ": members are not two pairable ops of one kind");
    }
    const int ngroup = (p->ops[k].flags >> HP_FLAG_GROUP_SHIFT) & HP_FLAG_GROUP_MASK;
    if (ngroup > 0) {
      if (ngroup >= HP_GROUP_MAX || (int)k - ngroup < 0 || (p->ops[k].flags & HP_FLAG_MEMBER))
        return fail("small-leaf group ending at op " + std::to_string(k) + ": bad length");
      for (int j = (int)k - ngroup; j <= (int)k; ++j) {
        const HpOp& m = p->ops[j];
        if ((j < (int)k && !(m.flags & HP_FLAG_MEMBER)) || !hp::groupable(m) ||
            (j < (int)k && ((m.flags >> HP_FLAG_GROUP_SHIFT) & HP_FLAG_GROUP_MASK)))
          return fail("small-leaf group ending at op " + std::to_string(k) + ": member " + std::to_string(j) + " is not a groupable member record");
      }
    }
    if (p->ops[k].op == HP_OP_HEADS) {
      const HpOp& g = p->ops[k];
      if (g.i[0] < 0 || g.i[1] <= 0 || g.i[0] + g.i[1] > (int)k) return fail("heads op " + std::to_string(k) + ": member range out of bounds");
      const char* w = "";
      if (!hp::check_heads(&p->ops[g.i[0]], g.i[1], g.i[2], &w)) return fail("heads op " + std::to_string(k) + ": " + w);
    }
    if (p->ops[k].op == HP_OP_WGRAD_GROUP) {
      const HpOp& g = p->ops[k];
      for (int j = g.i[0]; j < g.i[0] + g.i[1]; ++j) {
        const HpOp& m = p->ops[j];
        if (m.op != HP_OP_WGRAD_TAPS || !(m.flags & HP_FLAG_MEMBER) || !(m.flags & 1) || m.i[9] != g.i[2] ||
            (m.flags & (HP_CONV_BF16 | HP_CONV_BF16X3 | HP_FLAG_ACT_BF16)) != (p->ops[g.i[0]].flags & (HP_CONV_BF16 | HP_CONV_BF16X3 | HP_FLAG_ACT_BF16)))
          return fail("wgrad group member " + std::to_string(j) + " is not an atomic WGRAD_TAPS member with matching taps");
      }
    }
  }
  return 0;
}

int hp_program_run(HpProgram* p, int first, int count, void* stream) {
  if (!p) return fail("hp_program_run: null program");
  if (first < 0 || count < 0 || first + count > (int)p->ops.size()) return fail("hp_program_run: range out of bounds");
  hipStream_t s = (hipStream_t)stream;
  if (check_range(p, first, count, "hp_program_run")) return 1;
  if (ensure_groups(p)) return 1;
  for (int k = first; k < first + count; ++k) {
    hipError_t e = run_one(p, k, s);
    if (e != hipSuccess) {
      char buf[96];
      snprintf(buf, sizeof buf, "launch of op %d (opcode %d)", k, p->ops[k].op);
      return fail_hip(buf, e);
    }
  }
  return 0;
}

int hp_program_capture(HpProgram* p, int first, int count, int* seg) {
  if (!p || !seg) return fail("hp_program_capture: null argument");
  if (first < 0 || count <= 0 || first + count > (int)p->ops.size()) return fail("hp_program_capture: range out of bounds");
  if (check_range(p, first, count, "hp_program_capture")) return 1;
  hipError_t e;
  if (!p->capture_stream) {
    e = hipStreamCreateWithFlags(&p->capture_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail_hip("hipStreamCreate", e);
  }
  if (ensure_groups(p)) return 1;
  e = hipStreamBeginCapture(p->capture_stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return fail_hip("hipStreamBeginCapture", e);
  // a linear graph: multi-branch (fork/join) graphs were measured slower on this runtime
  int rc = 0;
  for (int k = first; k < first + count && rc == 0; ++k) {
    hipError_t le = run_one(p, k, p->capture_stream);
    if (le != hipSuccess) rc = fail_hip("capture launch", le);
  }
  hipGraph_t graph = nullptr;
  e = hipStreamEndCapture(p->capture_stream, &graph);
  if (rc != 0) { if (graph) hipGraphDestroy(graph); return 1; }
  if (e != hipSuccess) return fail_hip("hipStreamEndCapture", e);
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) { hipGraphDestroy(graph); return fail_hip("hipGraphInstantiate", e); }
  p->graphs.push_back(graph);
  p->segs.push_back(exec);
  *seg = (int)p->segs.size() - 1;
  return 0;
}

int hp_program_replay(HpProgram* p, int seg, void* stream) {
  if (!p || seg < 0 || seg >= (int)p->segs.size()) return fail("hp_program_replay: bad segment");
  hipError_t e = hipGraphLaunch(p->segs[seg], (hipStream_t)stream);
  if (e != hipSuccess) return fail_hip("hipGraphLaunch", e);
  return 0;
}

int hp_program_profile(HpProgram* p, int first, int count, void* stream, float* out_ms) {
  if (!p || !out_ms) return fail("hp_program_profile: null argument");
  if (first < 0 || count < 0 || first + count > (int)p->ops.size()) return fail("hp_program_profile: range out of bounds");
  hipStream_t s = (hipStream_t)stream;
  if (check_range(p, first, count, "hp_program_profile")) return 1;
  if (ensure_groups(p)) return 1;
  std::vector<hipEvent_t> ev(count + 1, nullptr);
  auto destroy = [&]() { for (auto& x : ev) if (x) hipEventDestroy(x); };
  for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) { destroy(); return fail("hipEventCreate failed"); }
  hipEventRecord(ev[0], s);
  for (int k = 0; k < count; ++k) {
    hipError_t e = run_one(p, first + k, s);
    if (e != hipSuccess) { hipStreamSynchronize(s); destroy(); return fail_hip("profile launch", e); }
    hipEventRecord(ev[k + 1], s);
  }
  hipError_t e = hipStreamSynchronize(s);
  if (e != hipSuccess) { destroy(); return fail_hip("hipStreamSynchronize", e); }
  for (int k = 0; k < count; ++k) hipEventElapsedTime(&out_ms[k], ev[k], ev[k + 1]);
  destroy();
  return 0;
}

int hp_run_op(const HpOp* op, void* const bases[HP_NUM_SPACES], void* stream) {
  if (!op || !bases) return fail("hp_run_op: null argument");
  if (op->op <= 0 || op->op >= HP_OP__COUNT) return fail("hp_run_op: unknown opcode");
  hipError_t e = dispatch(*op, bases, (hipStream_t)stream);
  if (e != hipSuccess) return fail_hip("hp_run_op launch", e);
  return 0;
}

}  // extern "C"
