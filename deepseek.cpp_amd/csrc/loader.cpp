// loader.cpp -- direct-to-HBM `.dseek` checkpoint loader.
//
// Replaces, for the HIP device, the reference's YALMData::from_directory (src/codec.cpp:333-365: every file of the
// directory in sorted order, `u64 header_len | JSON | data`, metadata from the first file) followed by the tensor
// walk of Model::Model / Block constructors (src/model.cpp:766-871) and Config::from_yalm (src/model.cpp:21-127).
// The reference mmaps the shards and keeps the weights in host memory; here every tensor's byte range is read with
// parallel preads into the context's pinned staging ring and copied to HBM while the next piece is being read
// (engine.cpp stage_copy); an expert-sharded rank reads only its own experts' bytes.  The on-disk format is unchanged.
#include "engine.h"

#include <dirent.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <vector>

namespace {

// ---- a JSON reader for the safetensors-style header: objects, arrays, strings, numbers, true / false / null ----
struct JVal {
  enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<JVal> arr;
  std::vector<std::pair<std::string, JVal>> obj;
  const JVal* get(const std::string& k) const {
    for (auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
};

struct JParser {
  const char* p;
  const char* end;
  bool ok = true;
  void ws() {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
  }
  bool lit(const char* s) {
    const size_t n = strlen(s);
    if ((size_t)(end - p) >= n && !memcmp(p, s, n)) { p += n; return true; }
    return false;
  }
  std::string string() {
    std::string out;
    if (p >= end || *p != '"') { ok = false; return out; }
    ++p;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (++p >= end) break;
        switch (*p) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {  // \uXXXX -> UTF-8 (BMP only; names and metadata are ASCII in practice)
            if (end - p < 5) { ok = false; return out; }
            unsigned cp = 0;
            for (int i = 1; i <= 4; ++i) {
              const char c = p[i];
              cp = cp * 16 + (c >= '0' && c <= '9' ? c - '0' : (c | 32) >= 'a' && (c | 32) <= 'f' ? (c | 32) - 'a' + 10 : (ok = false, 0));
            }
            p += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += *p;  // \" \\ \/
        }
        ++p;
      } else {
        out += *p++;
      }
    }
    if (p >= end) { ok = false; return out; }
    ++p;
    return out;
  }
  JVal value(int depth = 0) {
    JVal v;
    ws();
    if (p >= end || depth > 16) { ok = false; return v; }
    if (*p == '{') {
      v.kind = JVal::OBJ;
      ++p;
      ws();
      if (p < end && *p == '}') { ++p; return v; }
      while (ok) {
        ws();
        std::string k = string();
        ws();
        if (!ok || p >= end || *p != ':') { ok = false; break; }
        ++p;
        v.obj.emplace_back(std::move(k), value(depth + 1));
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == '}') { ++p; break; }
        ok = false;
      }
    } else if (*p == '[') {
      v.kind = JVal::ARR;
      ++p;
      ws();
      if (p < end && *p == ']') { ++p; return v; }
      while (ok) {
        v.arr.push_back(value(depth + 1));
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == ']') { ++p; break; }
        ok = false;
      }
    } else if (*p == '"') {
      v.kind = JVal::STR;
      v.str = string();
    } else if (lit("true")) { v.kind = JVal::BOOL; v.b = true; }
    else if (lit("false")) { v.kind = JVal::BOOL; }
    else if (lit("null")) { v.kind = JVal::NUL; }
    else {
      char* e = nullptr;
      v.kind = JVal::NUM;
      v.num = strtod(p, &e);
      if (e == p || e > end) ok = false;
      p = e;
    }
    return v;
  }
};

struct FileTensor {
  int file = -1;
  std::string dtype;
  int shape[4] = {0, 0, 0, 0};
  int ndim = 0;
  uint64_t off = 0, size = 0;  // absolute byte range in the file
};

struct Checkpoint {
  std::vector<std::string> files;
  std::vector<int> fds;
  std::map<std::string, std::string> meta;
  std::map<std::string, FileTensor> tensors;
  uint64_t tensor_bytes = 0;
  ~Checkpoint() {
    for (int fd : fds)
      if (fd >= 0) close(fd);
  }
};

size_t dtype_size(const std::string& d) {  // src/codec.cpp:108-122
  if (d == "F32" || d == "I32") return 4;
  if (d == "F16" || d == "BF16" || d == "I16") return 2;
  if (d == "F8_E5M2" || d == "F8_E4M3" || d == "I8" || d == "U8") return 1;
  return 0;
}

int read_shard(Checkpoint& ck, int fi, bool read_meta) {
  const std::string& fn = ck.files[fi];
  const int fd = open(fn.c_str(), O_RDONLY);
  if (fd < 0) DSK_FAIL(DSK_ERR_INVALID, "loader: cannot open %s", fn.c_str());
  ck.fds[fi] = fd;
  struct stat st;
  if (fstat(fd, &st) != 0) DSK_FAIL(DSK_ERR_INVALID, "loader: cannot stat %s", fn.c_str());
  const uint64_t size = (uint64_t)st.st_size;
  uint64_t jlen = 0;
  if (size < 8 || pread(fd, &jlen, 8, 0) != 8 || jlen == 0 || jlen > size - 8)  // src/codec.cpp:298-308
    DSK_FAIL(DSK_ERR_INVALID, "loader: %s has no valid header", fn.c_str());
  std::string js(jlen, '\0');
  for (uint64_t done = 0; done < jlen;) {
    const ssize_t r = pread(fd, &js[done], jlen - done, (off_t)(8 + done));
    if (r <= 0) DSK_FAIL(DSK_ERR_INVALID, "loader: short read of the header of %s", fn.c_str());
    done += (uint64_t)r;
  }
  posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);  // src/codec.cpp:290-293
  JParser P{js.data(), js.data() + js.size()};
  const JVal root = P.value();
  if (!P.ok || root.kind != JVal::OBJ) DSK_FAIL(DSK_ERR_INVALID, "loader: the header of %s is not a JSON object", fn.c_str());
  const uint64_t data0 = 8 + jlen, data_size = size - data0;
  for (auto& kv : root.obj) {
    if (kv.first == "__metadata__") {
      if (read_meta)
        for (auto& m : kv.second.obj)
          if (m.second.kind == JVal::STR) ck.meta[m.first] = m.second.str;  // all values are strings (convert.py:123-170)
      continue;
    }
    const JVal& v = kv.second;
    const JVal *dt = v.get("dtype"), *sh = v.get("shape"), *offs = v.get("data_offsets");
    FileTensor t;
    t.file = fi;
    if (!dt || dt->kind != JVal::STR || !(dtype_size(dt->str) > 0)) DSK_FAIL(DSK_ERR_INVALID, "loader: tensor %s: bad dtype", kv.first.c_str());
    t.dtype = dt->str;
    if (!sh || sh->kind != JVal::ARR) DSK_FAIL(DSK_ERR_INVALID, "loader: tensor %s: bad shape", kv.first.c_str());
    uint64_t numel = 1;
    for (size_t i = 0; i < sh->arr.size() && i < 4; ++i) {  // src/codec.cpp:135-146
      const double d = sh->arr[i].num;
      if (sh->arr[i].kind != JVal::NUM || d < 0 || d != (double)(int)d) DSK_FAIL(DSK_ERR_INVALID, "loader: tensor %s: bad shape", kv.first.c_str());
      t.shape[i] = (int)d;
      numel *= (uint64_t)(int)d;
      t.ndim = (int)i + 1;
    }
    if (!offs || offs->kind != JVal::ARR || offs->arr.size() != 2) DSK_FAIL(DSK_ERR_INVALID, "loader: tensor %s: bad data_offsets", kv.first.c_str());
    const double a = offs->arr[0].num, b = offs->arr[1].num;
    if (a < 0 || b <= a || b > (double)data_size) DSK_FAIL(DSK_ERR_INVALID, "loader: tensor %s: bad offsets", kv.first.c_str());  // :151-156
    t.off = data0 + (uint64_t)a;
    t.size = (uint64_t)b - (uint64_t)a;
    if (numel * dtype_size(t.dtype) != t.size) DSK_FAIL(DSK_ERR_INVALID, "loader: tensor %s: shape and size disagree", kv.first.c_str());  // :160-163
    ck.tensor_bytes += t.size;
    ck.tensors[kv.first] = t;  // a later shard overrides an earlier one, like the reference's map assignment
  }
  return DSK_OK;
}

int open_checkpoint(const char* dir, Checkpoint& ck) {
  DIR* d = opendir(dir);
  if (!d) DSK_FAIL(DSK_ERR_INVALID, "loader: cannot open directory %s", dir);
  while (dirent* e = readdir(d)) {
    const std::string n = e->d_name;
    if (n != "." && n != "..") ck.files.push_back(std::string(dir) + "/" + n);  // every entry, src/codec.cpp:343-350
  }
  closedir(d);
  if (ck.files.empty()) DSK_FAIL(DSK_ERR_INVALID, "loader: no files in %s", dir);
  std::sort(ck.files.begin(), ck.files.end());  // src/codec.cpp:358
  ck.fds.assign(ck.files.size(), -1);
  for (size_t i = 0; i < ck.files.size(); ++i) DSK_TRY(read_shard(ck, (int)i, i == 0));
  return DSK_OK;
}

// ---- Config::from_yalm (src/model.cpp:21-127), key by key ----
struct Meta {
  const std::map<std::string, std::string>& m;
  bool err = false;
  std::string missing;
  bool has(const char* k) const { return m.count(k) > 0; }
  const std::string& at(const char* k) {
    static const std::string empty;
    auto it = m.find(k);
    if (it == m.end()) {
      if (!err) missing = k;
      err = true;
      return empty;
    }
    return it->second;
  }
  int i(const char* k) { return atoi(at(k).c_str()); }                       // std::stoi
  float f(const char* k) { return strtof(at(k).c_str(), nullptr); }          // std::stof
  int i_or(const char* k, int dflt) { return has(k) ? i(k) : dflt; }
  float f_or(const char* k, float dflt) { return has(k) ? f(k) : dflt; }
  std::string s_or(const char* k, const char* dflt) { return has(k) ? at(k) : std::string(dflt); }
};

int config_from_meta(const std::map<std::string, std::string>& meta, int context, dsk_config* c) {
  memset(c, 0, sizeof *c);
  Meta M{meta};
  c->dim = M.i("dim");
  c->hidden_dim = M.i("hidden_dim");
  c->n_layers = M.i("n_layers");
  c->n_heads = M.i("n_heads");
  c->vocab_size = M.i("vocab_size");
  c->n_shared_experts = M.i_or("n_shared_experts", 0);
  c->n_routed_experts = M.i_or("n_routed_experts", 0);
  c->n_active_routed = M.i_or("n_active_routed", 0);
  c->moe_intermediate_size = M.i_or("moe_intermediate_size", 0);
  c->routed_scaling_factor = M.f_or("routed_scaling_factor", 1.0f);
  c->n_group = M.i_or("n_group", 1);
  c->norm_topk_prob = M.has("norm_topk_prob") && M.at("norm_topk_prob") == "True";
  const std::string scoring = M.s_or("scoring_func", "softmax");
  c->scoring_func = scoring == "sigmoid" ? DSK_SCORE_SIGMOID : DSK_SCORE_SOFTMAX;  // unknown: softmax, like the reference
  c->topk_group = M.i_or("topk_group", 0);
  const std::string topk = M.s_or("topk_method", "");
  if (topk == "noaux_tc") DSK_FAIL(DSK_ERR_UNSUPPORTED, "loader: topk_method noaux_tc (the reference asserts here too, src/model.cpp:52-54)");
  c->topk_method = topk == "group_limited_greedy" ? DSK_TOPK_GROUP_LIMITED_GREEDY : DSK_TOPK_GREEDY;  // unknown: greedy
  c->has_moegate_bias = M.at("arch") == "DeepseekV3ForCausalLM";
  c->use_mla = M.has("use_mla") ? (M.i("use_mla") != 0) : 0;
  c->kv_lora_rank = M.i_or("kv_lora_rank", 0);
  c->q_lora_rank = M.i_or("q_lora_rank", 0);
  c->qk_nope_head_dim = M.i_or("qk_nope_head_dim", 0);
  c->qk_rope_head_dim = M.i_or("qk_rope_head_dim", 0);
  c->v_head_dim = M.i_or("v_head_dim", 0);
  c->max_seq_len = M.i("max_seq_len");
  if (context) c->max_seq_len = std::min(c->max_seq_len, context);
  c->rope_theta = M.f("rope_theta");
  c->norm_eps = strtof(M.s_or("norm_eps", "1e-5").c_str(), nullptr);
  c->act = M.s_or("act_type", "gelu") == "silu" ? DSK_ACT_SILU : DSK_ACT_GELU;  // unknown: gelu
  c->first_k_dense_replace = M.i_or("first_k_dense_replace", 0);
  const std::string q = M.at("quant");
  if (q == "fp32") c->weight_quant = DSK_QUANT_F32;
  else if (q == "fp16") c->weight_quant = DSK_QUANT_F16;
  else if (q == "f8e5m2") c->weight_quant = DSK_QUANT_F8E5M2;
  else if (q == "q2_k") c->weight_quant = DSK_QUANT_Q2_K;
  else if (q == "q3_k") c->weight_quant = DSK_QUANT_Q3_K;
  else if (!M.err) DSK_FAIL(DSK_ERR_UNSUPPORTED, "loader: unsupported quant '%s'", q.c_str());
  if (M.has("quantization_block_size_0")) {
    c->block_size[0] = M.i("quantization_block_size_0");
    c->block_size[1] = M.i("quantization_block_size_1");
  }
  // the RoPE-scaling keys are required by the reference even though the path only uses the ring modulus
  M.at("rope_scaling_beta_fast"); M.at("rope_scaling_beta_slow"); M.at("rope_scaling_factor");
  M.at("rope_scaling_mscale"); M.at("rope_scaling_mscale_all_dim");
  c->rs_original_max_position_embeddings = M.i("rope_scaling_original_max_position_embeddings");
  if (M.err) DSK_FAIL(DSK_ERR_INVALID, "loader: metadata key '%s' is missing", M.missing.c_str());
  return DSK_OK;
}

const char* quant_dtype(int quant) {  // quant_to_codec_dtype, src/codec.cpp:60-77
  switch (quant) {
    case DSK_QUANT_F32: return "F32";
    case DSK_QUANT_F16: return "F16";
    case DSK_QUANT_F8E5M2: return "F8_E5M2";
    default: return "U8";
  }
}

struct RoleName { int role; const char* name; };
const RoleName LAYER_NAMES[] = {
    {DSK_ROLE_ATTN_NORM, "attn.norm"},       {DSK_ROLE_Q_A_NORM, "attn.q_a_norm"},   {DSK_ROLE_KV_A_NORM, "attn.kv_a_norm"},
    {DSK_ROLE_FFN_NORM, "mlp.norm"},         {DSK_ROLE_WQ, "attn.wq"},               {DSK_ROLE_WQ_A, "attn.wq_a"},
    {DSK_ROLE_WQ_B, "attn.wq_b"},            {DSK_ROLE_WKV_A, "attn.wkv_a"},         {DSK_ROLE_WKV_B, "attn.wkv_b"},
    {DSK_ROLE_WO, "attn.wo"},                {DSK_ROLE_WC, "attn.wc"},               {DSK_ROLE_WQ_ROPE_B, "attn.wq_rope_b"},
    {DSK_ROLE_WV_B, "attn.wv_b"},            {DSK_ROLE_W1, "mlp.w1"},                {DSK_ROLE_W2, "mlp.w2"},
    {DSK_ROLE_W3, "mlp.w3"},                 {DSK_ROLE_SHARED_W1, "shared_mlp.w1"},  {DSK_ROLE_SHARED_W2, "shared_mlp.w2"},
    {DSK_ROLE_SHARED_W3, "shared_mlp.w3"},   {DSK_ROLE_MOEGATE, "moegate"}};

struct Walk {
  dsk_model* m;
  Checkpoint& ck;
  const dsk_config& c;
  bool planes = false;  // the checkpoint stores K-quant tensors in the engine's plane layout
  int n_bound = 0;
  uint64_t file_bytes = 0;

  // one tensor the reference constructors fetch by name; `required` mirrors their asserts
  int bind(const std::string& name, int role, int layer, bool required) {
    auto it = ck.tensors.find(name);
    if (it == ck.tensors.end()) {
      if (required) DSK_FAIL(DSK_ERR_INVALID, "loader: tensor %s is missing", name.c_str());
      return DSK_OK;
    }
    const FileTensor& ft = it->second;
    const bool is_scale = role >= DSK_ROLE_SCALE;
    if (planes && !is_scale && is_kq(role_shape(m, role, layer).quant) && role_shape(m, role, layer).ok)
      return bind_plane_set(name, role, layer);
    const RoleShape rs = role_shape(m, is_scale ? role - DSK_ROLE_SCALE : role, layer);
    if (!rs.ok) return DSK_OK;  // present in the file, not part of this configuration: the reference never asks for it
    const int quant = is_scale ? DSK_QUANT_F32 : rs.quant;
    if (ft.dtype != quant_dtype(quant)) DSK_FAIL(DSK_ERR_INVALID, "loader: tensor %s has dtype %s, expected %s", name.c_str(), ft.dtype.c_str(), quant_dtype(quant));
    int32_t shape[4] = {0, 0, 0, 0};
    if (!is_scale) {
      // the logical shape the constructors pass to check_tensor (src/model.cpp:129-150); K-quant tensors are stored as
      // bytes, so for them only the byte count is checked (src/codec.cpp:170-207), like in the reference
      if (rs.e > 0) { shape[0] = rs.e; shape[1] = rs.rows; shape[2] = rs.n; }
      else if (rs.rows == 1 && rs.quant == DSK_QUANT_F32 && role != DSK_ROLE_MOEGATE) shape[0] = rs.n;
      else { shape[0] = rs.rows; shape[1] = rs.n; }
      if (!is_kq(quant))
        for (int i = 0; i < 4; ++i)
          if (ft.shape[i] != shape[i])
            DSK_FAIL(DSK_ERR_INVALID, "loader: tensor %s has shape [%d,%d,%d,%d], expected [%d,%d,%d,%d]", name.c_str(), ft.shape[0], ft.shape[1],
                     ft.shape[2], ft.shape[3], shape[0], shape[1], shape[2], shape[3]);
    }
    HostSrc src;
    src.fd = ck.fds[ft.file];
    src.off = ft.off;
    DSK_TRY(bind_src(m, role, layer, quant, shape, src, ft.size));
    ++n_bound;
    file_bytes += ft.size;
    return DSK_OK;
  }
  // plane layout (metadata gpu_layout = planes-v1, tools/repack.py): "<name>" is a 1-byte marker, the data sit in
  // "<name>.qs" / ".sc" / ".hm" (Q3_K) / ".dm"
  int bind_plane_set(const std::string& name, int role, int layer) {
    const RoleShape rs = role_shape(m, role, layer);
    static const char* SUF[4] = {".qs", ".sc", ".hm", ".dm"};
    HostSrc src[4];
    size_t bytes[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      if (i == 2 && rs.quant != DSK_QUANT_Q3_K) continue;
      auto it = ck.tensors.find(name + SUF[i]);
      if (it == ck.tensors.end()) DSK_FAIL(DSK_ERR_INVALID, "loader: plane %s%s is missing", name.c_str(), SUF[i]);
      if (it->second.dtype != "U8") DSK_FAIL(DSK_ERR_INVALID, "loader: plane %s%s is not U8", name.c_str(), SUF[i]);
      src[i].fd = ck.fds[it->second.file];
      src[i].off = it->second.off;
      bytes[i] = it->second.size;
      file_bytes += it->second.size;
    }
    DSK_TRY(bind_planes(m, role, layer, rs.quant, src, bytes));
    ++n_bound;
    return DSK_OK;
  }
  int weight(const std::string& base, int role, int layer, bool required) {
    DSK_TRY(bind(base + ".weight", role, layer, required));
    // need_weight_scales (src/model.cpp:191): an optional weight that IS present needs its scale too (model.output,
    // src/model.cpp:857-864)
    const bool have = ck.tensors.find(base + ".weight") != ck.tensors.end();
    if (c.weight_quant == DSK_QUANT_F8E5M2 && role_shape(m, role, layer).quant == DSK_QUANT_F8E5M2)
      DSK_TRY(bind(base + ".scale", role + DSK_ROLE_SCALE, layer, required || have));
    return DSK_OK;
  }
  int all() {
    DSK_TRY(weight("model.embed", DSK_ROLE_EMBED, -1, true));
    DSK_TRY(bind("model.norm.weight", DSK_ROLE_FINAL_NORM, -1, true));
    DSK_TRY(weight("model.output", DSK_ROLE_OUTPUT, -1, false));  // absent => tied (src/model.cpp:852-856)
    for (int l = 0; l < c.n_layers; ++l) {
      const std::string p = "model.layers." + std::to_string(l) + ".";
      for (const RoleName& rn : LAYER_NAMES) {
        if (!role_shape(m, rn.role, l).ok) continue;
        DSK_TRY(weight(p + rn.name, rn.role, l, true));
      }
      if (role_shape(m, DSK_ROLE_MOEGATE_BIAS, l).ok) DSK_TRY(bind(p + "moegate.bias", DSK_ROLE_MOEGATE_BIAS, l, true));
    }
    return DSK_OK;
  }
};

}  // namespace

extern "C" int dsk_dseek_read_config(const char* dir, int context, dsk_config* out, int32_t* n_files, int32_t* n_tensors, uint64_t* tensor_bytes) {
  if (!dir || !out) DSK_FAIL(DSK_ERR_INVALID, "dseek_read_config: null argument");
  Checkpoint ck;
  DSK_TRY(open_checkpoint(dir, ck));
  DSK_TRY(config_from_meta(ck.meta, context, out));
  if (n_files) *n_files = (int32_t)ck.files.size();
  if (n_tensors) *n_tensors = (int32_t)ck.tensors.size();
  if (tensor_bytes) *tensor_bytes = ck.tensor_bytes;
  return DSK_OK;
}

extern "C" int dsk_model_load_dseek(dsk_ctx* ctx, const char* dir, int context, dsk_model** out, dsk_load_stats* stats) {
  return dsk_model_load_dseek_opts(ctx, dir, context, nullptr, out, stats);
}

// ... with model options ("key=value,key=value": dsk_model_set_option between create and the first bind), e.g. "q2k_tiles=2"
// for a checkpoint whose prompts should go through the batched dsk_hydrate path
extern "C" int dsk_model_load_dseek_opts(dsk_ctx* ctx, const char* dir, int context, const char* options, dsk_model** out, dsk_load_stats* stats) {
  if (!ctx || !dir || !out) DSK_FAIL(DSK_ERR_INVALID, "load_dseek: null argument");
  *out = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  Checkpoint ck;
  DSK_TRY(open_checkpoint(dir, ck));
  dsk_config c;
  DSK_TRY(config_from_meta(ck.meta, context, &c));
  dsk_model* m = nullptr;
  DSK_TRY(dsk_model_create(ctx, &c, &m));
  for (const char* p = options; p && *p;) {  // key=value[,key=value...]
    const char* e = strchr(p, ',');
    const std::string kv(p, e ? (size_t)(e - p) : strlen(p));
    const size_t eq = kv.find('=');
    int r = DSK_ERR_INVALID;
    if (eq != std::string::npos && eq > 0 && eq + 1 < kv.size()) r = dsk_model_set_option(m, kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1));
    else dsk_set_error(DSK_ERR_INVALID, "load_dseek: malformed option '%s' (key=value expected)", kv.c_str());
    if (r != DSK_OK) {
      const std::string keep = dsk_last_error();
      dsk_model_destroy(m);
      DSK_FAIL(r, "%s", keep.c_str());
    }
    p = e ? e + 1 : nullptr;
  }
  const double staged0 = ctx->staged_bytes, fill0 = ctx->staged_fill_s;
  Walk W{m, ck, m->c};
  {
    auto it = ck.meta.find("gpu_layout");
    if (it != ck.meta.end()) {
      if (it->second != "planes-v1") {
        dsk_model_destroy(m);
        DSK_FAIL(DSK_ERR_UNSUPPORTED, "loader: unknown gpu_layout '%s'", it->second.c_str());
      }
      W.planes = true;
    }
  }
  int r = W.all();
  if (r == DSK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) r = DSK_ERR_HIP;
  if (r == DSK_OK) r = dsk_model_finalize(m);
  if (r != DSK_OK) {
    const std::string keep = dsk_last_error();
    dsk_model_destroy(m);
    DSK_FAIL(r, "%s", keep.c_str());
  }
  if (stats) {
    stats->n_files = (int32_t)ck.files.size();
    stats->n_tensors = W.n_bound;
    stats->file_bytes = W.file_bytes;
    stats->staged_bytes = (uint64_t)(ctx->staged_bytes - staged0);
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats->read_seconds = ctx->staged_fill_s - fill0;
  }
  *out = m;
  return DSK_OK;
}
