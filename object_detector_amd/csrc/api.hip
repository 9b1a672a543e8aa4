// Context, error reporting and the native forward plan of libodhip.so.
#include <stddef.h>
#include <string.h>

#include <string>
#include <vector>

#include <mutex>

#include "common.h"

static thread_local char g_err[1024] = "";

void od_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* od_last_error(void) { return g_err; }
extern "C" int od_version(void) { return 100; }

extern "C" int od_ctx_create(int device, od_ctx** out) {
  OD_REQUIRE(out, "od_ctx_create: null out");
  OD_CHECK_HIP(hipSetDevice(device));
  od_ctx* c = new od_ctx();
  c->device = device;
  c->zero_page = nullptr;
  c->ones = nullptr;
  hipDeviceProp_t prop;
  OD_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  c->num_cu = prop.multiProcessorCount;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    od_set_error("od_ctx_create: device %d is %s; libodhip.so is built for gfx950 (MI355X) only", device,
                 prop.gcnArchName);
    delete c;
    return OD_ERR_INVALID;
  }
  OD_CHECK_HIP(hipMalloc(&c->zero_page, 8192));
  OD_CHECK_HIP(hipMemset(c->zero_page, 0, 8192));
  {
    std::vector<float> one(2048, 1.0f);
    OD_CHECK_HIP(hipMalloc((void**)&c->ones, 8192));
    OD_CHECK_HIP(hipMemcpy(c->ones, one.data(), 8192, hipMemcpyHostToDevice));
  }
  *out = c;
  return OD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// ABI self-description (od_sizeof / od_offsetof / od_struct_fields)
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct FieldInfo {
  const char* name;
  long offset;
};
struct StructInfo {
  const char* name;
  long size;
  std::vector<FieldInfo> fields;
};
#define OD_F(S, f) {#f, (long)offsetof(S, f)}
const std::vector<StructInfo>& od_struct_table() {
  static const std::vector<StructInfo> t = {
      {"od_conv_desc", sizeof(od_conv_desc),
       {OD_F(od_conv_desc, x), OD_F(od_conv_desc, w), OD_F(od_conv_desc, scale), OD_F(od_conv_desc, bias),
        OD_F(od_conv_desc, res), OD_F(od_conv_desc, out), OD_F(od_conv_desc, B), OD_F(od_conv_desc, H),
        OD_F(od_conv_desc, W), OD_F(od_conv_desc, Cin), OD_F(od_conv_desc, Cout), OD_F(od_conv_desc, ksize),
        OD_F(od_conv_desc, stride), OD_F(od_conv_desc, act), OD_F(od_conv_desc, alpha), OD_F(od_conv_desc, res_mode),
        OD_F(od_conv_desc, out_dtype), OD_F(od_conv_desc, out_batch_stride), OD_F(od_conv_desc, out_pix_stride),
        OD_F(od_conv_desc, tile_cfg), OD_F(od_conv_desc, transposed), OD_F(od_conv_desc, splitk),
        OD_F(od_conv_desc, splitk_workspace), OD_F(od_conv_desc, splitk_workspace_bytes), OD_F(od_conv_desc, bn_partials),
        OD_F(od_conv_desc, bn_partials_bytes), OD_F(od_conv_desc, w2), OD_F(od_conv_desc, scale2), OD_F(od_conv_desc, bias2),
        OD_F(od_conv_desc, out2), OD_F(od_conv_desc, Cout2), OD_F(od_conv_desc, act2), OD_F(od_conv_desc, alpha2),
        OD_F(od_conv_desc, pad2_), OD_F(od_conv_desc, nseg), OD_F(od_conv_desc, pad3_), OD_F(od_conv_desc, seg_x),
        OD_F(od_conv_desc, seg_out), OD_F(od_conv_desc, seg_H), OD_F(od_conv_desc, seg_W)}},
      {"od_bneck_desc", sizeof(od_bneck_desc),
       {OD_F(od_bneck_desc, x), OD_F(od_bneck_desc, w1), OD_F(od_bneck_desc, scale1), OD_F(od_bneck_desc, bias1),
        OD_F(od_bneck_desc, w3), OD_F(od_bneck_desc, scale3), OD_F(od_bneck_desc, bias3), OD_F(od_bneck_desc, out),
        OD_F(od_bneck_desc, B), OD_F(od_bneck_desc, H), OD_F(od_bneck_desc, W), OD_F(od_bneck_desc, C),
        OD_F(od_bneck_desc, act), OD_F(od_bneck_desc, alpha)}},
      {"od_stem_desc", sizeof(od_stem_desc),
       {OD_F(od_stem_desc, x), OD_F(od_stem_desc, w0), OD_F(od_stem_desc, scale0), OD_F(od_stem_desc, bias0),
        OD_F(od_stem_desc, w3), OD_F(od_stem_desc, scale3), OD_F(od_stem_desc, bias3), OD_F(od_stem_desc, out),
        OD_F(od_stem_desc, B), OD_F(od_stem_desc, H), OD_F(od_stem_desc, W), OD_F(od_stem_desc, act),
        OD_F(od_stem_desc, alpha)}},
      {"od_wgrad_red", sizeof(od_wgrad_red),
       {OD_F(od_wgrad_red, dw_offset), OD_F(od_wgrad_red, count), OD_F(od_wgrad_red, slabs), OD_F(od_wgrad_red, nslabs),
        OD_F(od_wgrad_red, pad_)}},
      {"od_sgd_seg", sizeof(od_sgd_seg),
       {OD_F(od_sgd_seg, offset), OD_F(od_sgd_seg, count), OD_F(od_sgd_seg, lr), OD_F(od_sgd_seg, weight_decay)}},
      {"od_pack_layer", sizeof(od_pack_layer),
       {OD_F(od_pack_layer, w_offset), OD_F(od_pack_layer, w_fwd), OD_F(od_pack_layer, w_bwd), OD_F(od_pack_layer, Cout),
        OD_F(od_pack_layer, Cin), OD_F(od_pack_layer, ksize), OD_F(od_pack_layer, pad_)}},
      {"od_aug_params", sizeof(od_aug_params),
       {OD_F(od_aug_params, src_offset), OD_F(od_aug_params, src_h), OD_F(od_aug_params, src_w),
        OD_F(od_aug_params, crop_x1), OD_F(od_aug_params, crop_y1), OD_F(od_aug_params, crop_x2),
        OD_F(od_aug_params, crop_y2), OD_F(od_aug_params, flip), OD_F(od_aug_params, brightness),
        OD_F(od_aug_params, contrast), OD_F(od_aug_params, saturation), OD_F(od_aug_params, n_erase),
        OD_F(od_aug_params, erase), OD_F(od_aug_params, erase_rgb)}},
      {"od_plan_op", sizeof(od_plan_op),
       {OD_F(od_plan_op, kind), OD_F(od_plan_op, pad_), OD_F(od_plan_op, conv), OD_F(od_plan_op, bneck),
        OD_F(od_plan_op, stem), OD_F(od_plan_op, wide)}},
      {"od_wide_desc", sizeof(od_wide_desc),
       {OD_F(od_wide_desc, y), OD_F(od_wide_desc, res), OD_F(od_wide_desc, out32), OD_F(od_wide_desc, out16),
        OD_F(od_wide_desc, out_hilo), OD_F(od_wide_desc, M), OD_F(od_wide_desc, C), OD_F(od_wide_desc, res_f32),
        OD_F(od_wide_desc, res_up2), OD_F(od_wide_desc, H), OD_F(od_wide_desc, W), OD_F(od_wide_desc, pad_)}},
  };
  return t;
}
#undef OD_F
const StructInfo* od_find_struct(const char* name) {
  if (!name) return nullptr;
  for (const StructInfo& s : od_struct_table())
    if (strcmp(s.name, name) == 0) return &s;
  return nullptr;
}
}  // namespace

extern "C" long od_sizeof(const char* struct_name) {
  const StructInfo* s = od_find_struct(struct_name);
  return s ? s->size : -1;
}

extern "C" long od_offsetof(const char* struct_name, const char* field_name) {
  const StructInfo* s = od_find_struct(struct_name);
  if (!s || !field_name) return -1;
  for (const FieldInfo& f : s->fields)
    if (strcmp(f.name, field_name) == 0) return f.offset;
  return -1;
}

extern "C" int od_struct_fields(const char* struct_name, char* buf, int buf_bytes) {
  const StructInfo* s = od_find_struct(struct_name);
  if (!s || !buf) return -1;
  std::string out;
  for (const FieldInfo& f : s->fields) {
    if (!out.empty()) out += ",";
    out += f.name;
  }
  if ((int)out.size() + 1 > buf_bytes) return -1;
  memcpy(buf, out.c_str(), out.size() + 1);
  return (int)s->fields.size();
}

int od_ensure_lds(od_ctx* ctx, const void* fn, size_t lds) {
  // the one piece of mutable context state: a data-generator thread (od_gen prefetch) may issue kernels beside the thread
  // that drives the training step
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  auto it = ctx->lds_attr.find(fn);
  if (it != ctx->lds_attr.end() && it->second >= lds) return OD_OK;
  OD_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ctx->lds_attr[fn] = lds;
  return OD_OK;
}

extern "C" int od_ctx_destroy(od_ctx* ctx) {
  if (!ctx) return OD_OK;
  if (ctx->zero_page) (void)hipFree(ctx->zero_page);
  if (ctx->ones) (void)hipFree(ctx->ones);
  delete ctx;
  return OD_OK;
}

// A HIP stream confined to a subset of the CUs (hipExtStreamCreateWithCUMask): the training step runs its weight-gradient
// chain on one beside the dz -> dx chain, so that the two chains stop evicting each other's tiles from every CU.
// cu_bits: bit i of word i / 32 = CU i enabled.
extern "C" int od_stream_create_cu_mask(od_ctx* ctx, const uint32_t* cu_bits, int n_words, void** out) {
  OD_REQUIRE(ctx && cu_bits && n_words > 0 && out, "od_stream_create_cu_mask: bad argument");
  OD_CHECK_HIP(hipSetDevice(ctx->device));
  hipStream_t s = nullptr;
  OD_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_bits));
  *out = (void*)s;
  return OD_OK;
}

extern "C" int od_stream_destroy(od_ctx* ctx, void* stream) {
  OD_REQUIRE(ctx && stream, "od_stream_destroy: bad argument");
  OD_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
  return OD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Forward plan: the layer list of one network, launched from C++ (eager or as a replayed hipGraph).
// ---------------------------------------------------------------------------------------------------------------
struct od_plan {
  od_ctx* ctx;
  std::vector<od_plan_op> ops;
  std::vector<const char*> names;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

static int run_op(od_plan* pl, int i, hipStream_t s) {
  const od_plan_op& op = pl->ops[i];
  if (op.kind == OD_OP_CONV) return od_conv2d_fwd_impl(pl->ctx, &op.conv, s, nullptr, false);
  if (op.kind == OD_OP_CONV_FIRST) {
    const od_conv_desc& c = op.conv;
    return od_conv_first_fwd(pl->ctx, (const uint8_t*)c.x, c.w, c.scale, c.bias, c.out, c.B, c.H, c.W, c.Cout, c.act,
                             c.alpha, s);
  }
  if (op.kind == OD_OP_BNECK) return od_bottleneck_fwd(pl->ctx, &op.bneck, s);
  if (op.kind == OD_OP_STEM) return od_stem_fwd(pl->ctx, &op.stem, s);
  if (op.kind == OD_OP_WIDE) return od_wide_add(pl->ctx, &op.wide, s);
  od_set_error("od_plan: unknown op kind %d at %d", op.kind, i);
  return OD_ERR_INVALID;
}

extern "C" int od_plan_create(od_ctx* ctx, const od_plan_op* ops, int n_ops, od_plan** out) {
  OD_REQUIRE(ctx && ops && n_ops > 0 && out, "od_plan_create: bad args");
  od_plan* pl = new od_plan();
  pl->ctx = ctx;
  pl->ops.assign(ops, ops + n_ops);
  pl->names.resize(n_ops, "");
  for (int i = 0; i < n_ops; ++i) {
    if (ops[i].kind == OD_OP_CONV) {
      const char* nm = nullptr;
      int rc = od_conv2d_fwd_impl(ctx, &ops[i].conv, nullptr, &nm, true);  // validates the descriptor
      if (rc != OD_OK) {
        delete pl;
        return rc;
      }
      pl->names[i] = nm;
    } else if (ops[i].kind == OD_OP_CONV_FIRST) {
      pl->names[i] = od_conv_first_kernel_name();
    } else if (ops[i].kind == OD_OP_STEM) {
      if (!od_stem_supported(ops[i].stem.H, ops[i].stem.W)) {
        od_set_error("od_plan_create: op %d: fused stem needs H, W multiples of 32 (got %dx%d)", i, ops[i].stem.H, ops[i].stem.W);
        delete pl;
        return OD_ERR_INVALID;
      }
      pl->names[i] = od_stem_kernel_name();
    } else if (ops[i].kind == OD_OP_BNECK) {
      if (!od_bottleneck_supported(ops[i].bneck.H, ops[i].bneck.W, ops[i].bneck.C)) {
        od_set_error("od_plan_create: op %d: fused block unsupported for C=%d, %dx%d", i, ops[i].bneck.C, ops[i].bneck.H,
                     ops[i].bneck.W);
        delete pl;
        return OD_ERR_INVALID;
      }
      pl->names[i] = od_bottleneck_kernel_name(ops[i].bneck.C);
    } else if (ops[i].kind == OD_OP_WIDE) {
      pl->names[i] = "od_wide_add_k";
    } else {
      od_set_error("od_plan_create: unknown op kind %d at %d", ops[i].kind, i);
      delete pl;
      return OD_ERR_INVALID;
    }
  }
  *out = pl;
  return OD_OK;
}

extern "C" int od_plan_run(od_plan* pl, void* stream) {
  OD_REQUIRE(pl, "od_plan_run: null plan");
  for (size_t i = 0; i < pl->ops.size(); ++i) {
    int rc = run_op(pl, (int)i, (hipStream_t)stream);
    if (rc != OD_OK) return rc;
  }
  return OD_OK;
}

extern "C" int od_plan_capture(od_plan* pl, void* stream) {
  OD_REQUIRE(pl, "od_plan_capture: null plan");
  hipStream_t s = (hipStream_t)stream;
  OD_REQUIRE(s != nullptr, "od_plan_capture: needs a non-default stream");
  // warm every kernel once outside capture (function attributes are set lazily on first launch)
  int rc = od_plan_run(pl, stream);
  if (rc != OD_OK) return rc;
  OD_CHECK_HIP(hipStreamSynchronize(s));
  if (pl->exec) {
    (void)hipGraphExecDestroy(pl->exec);
    pl->exec = nullptr;
  }
  if (pl->graph) {
    (void)hipGraphDestroy(pl->graph);
    pl->graph = nullptr;
  }
  OD_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  rc = od_plan_run(pl, stream);
  hipError_t e = hipStreamEndCapture(s, &pl->graph);
  if (rc != OD_OK) return rc;
  if (e != hipSuccess) {
    od_set_error("od_plan_capture: hipStreamEndCapture -> %s", hipGetErrorString(e));
    return OD_ERR_HIP;
  }
  OD_CHECK_HIP(hipGraphInstantiate(&pl->exec, pl->graph, nullptr, nullptr, 0));
  return OD_OK;
}

extern "C" int od_plan_replay(od_plan* pl, void* stream) {
  OD_REQUIRE(pl && pl->exec, "od_plan_replay: plan not captured");
  OD_CHECK_HIP(hipGraphLaunch(pl->exec, (hipStream_t)stream));
  return OD_OK;
}

extern "C" int od_plan_destroy(od_plan* pl) {
  if (!pl) return OD_OK;
  if (pl->exec) (void)hipGraphExecDestroy(pl->exec);
  if (pl->graph) (void)hipGraphDestroy(pl->graph);
  delete pl;
  return OD_OK;
}

extern "C" int od_plan_time_ops(od_plan* pl, void* stream, float* ms, int n_ops) {
  OD_REQUIRE(pl && ms && n_ops == (int)pl->ops.size(), "od_plan_time_ops: bad args");
  hipStream_t s = (hipStream_t)stream;
  std::vector<hipEvent_t> ev(n_ops + 1);
  for (auto& e : ev) OD_CHECK_HIP(hipEventCreate(&e));
  OD_CHECK_HIP(hipEventRecord(ev[0], s));
  int rc = OD_OK;
  for (int i = 0; i < n_ops && rc == OD_OK; ++i) {
    rc = run_op(pl, i, s);
    if (rc == OD_OK) OD_CHECK_HIP(hipEventRecord(ev[i + 1], s));
  }
  if (rc == OD_OK) {
    OD_CHECK_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n_ops; ++i) OD_CHECK_HIP(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

extern "C" const char* od_plan_op_kernel_name(od_plan* pl, int i) {
  if (!pl || i < 0 || i >= (int)pl->names.size()) return "";
  return pl->names[i];
}
