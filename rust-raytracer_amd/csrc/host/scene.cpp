// scene.cpp — host plumbing of librt_host.so: the JSON scene schema, Camera::new,
// find_lights, PNG output.  C++ stand-in for the Rust host code that stays on the CPU
// (reference main.rs:7-20, config.rs:20-75, camera.rs:9-77, sphere.rs:18-23,
// materials.rs:18-42/56-76/97-103/131-134/201-234, raytracer.rs:33-42/220-229).
// No path-tracing arithmetic lives here; that is librt_hip.so.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <initializer_list>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rt_abi.h"
#include "json.hpp"

extern "C" const char* rt_jpeg_last_error(void);

namespace {

thread_local std::string g_err;
int set_err(int code, const std::string& m) { g_err = m; return code; }

struct SchemaError { std::string msg; };
[[noreturn]] void bad(const std::string& m) { throw SchemaError{m}; }
struct UnsupportedError { std::string msg; };  // the document is what serde would accept, this build cannot hold it (RT_ERR_UNSUPPORTED)

using rtjson::Value;

double as_f64(const Value& v, const char* what) {
  if (v.kind != Value::Number) bad(std::string("expected number for ") + what);
  const double x = std::strtod(v.text.c_str(), nullptr);
  // serde_json rejects a literal beyond f64's range ("number out of range"); it never yields an infinity
  if (!std::isfinite(x)) bad(std::string("number out of range for ") + what);
  return x;
}
float as_f32(const Value& v, const char* what) {
  // serde_json parses the literal as f64 and casts (`as f32`)
  return static_cast<float>(as_f64(v, what));
}
uint64_t as_u64(const Value& v, const char* what, uint64_t max) {
  if (v.kind != Value::Number) bad(std::string("expected unsigned integer for ") + what);
  const std::string& t = v.text;
  if (t.empty() || t.find_first_not_of("0123456789") != std::string::npos)
    bad(std::string("invalid type: expected unsigned integer for ") + what);
  errno = 0;
  unsigned long long x = std::strtoull(t.c_str(), nullptr, 10);
  if (errno == ERANGE || x > max) bad(std::string("integer out of range for ") + what);
  return x;
}
// A serde-derive struct (point3d.rs:10-15, camera.rs:29-36, sphere.rs:18-23, materials.rs:56-57/71-76/97-103/131-134/201-211,
// config.rs:20-28/66-75): serde_json's deserialize_struct takes it as a MAP — unknown keys ignored, a known key twice is the
// error "duplicate field", a missing one "missing field" (an Option field: None) — or as a SEQUENCE of exactly its fields in
// declaration order ("center":[0,0,0], "Light":[]).  out[i] = the value of keys[i]; nullptr only for a missing optional field.
void struct_fields(const Value& v, const char* name, std::initializer_list<const char*> keys, const Value** out,
                   std::initializer_list<bool> optional = {}) {
  const size_t n = keys.size();
  for (size_t i = 0; i < n; ++i) out[i] = nullptr;
  if (v.kind == Value::Array) {
    if (v.items.size() != n)
      bad("invalid length " + std::to_string(v.items.size()) + ", expected struct " + name + " with " + std::to_string(n) + " element" + (n == 1 ? "" : "s"));
    for (size_t i = 0; i < n; ++i) out[i] = v.items[i].get();
    return;
  }
  if (v.kind != Value::Object) bad(std::string("invalid type: expected struct ") + name);
  for (auto& m : v.members) {
    size_t i = 0;
    for (const char* k : keys) {
      if (m.first == k) {
        if (out[i]) bad(std::string("duplicate field `") + k + "`");
        out[i] = m.second.get();
        break;
      }
      ++i;
    }
  }
  size_t i = 0;
  for (const char* k : keys) {
    const bool opt = i < optional.size() && *(optional.begin() + i);
    if (!out[i] && !opt) bad(std::string("missing field `") + k + "` in " + name);
    ++i;
  }
}
void as_point(const Value& v, const char* what, double out[3]) {
  const Value* f[3];
  struct_fields(v, "Point3D", {"x", "y", "z"}, f);
  for (int i = 0; i < 3; ++i) out[i] = as_f64(*f[i], what);
}
// usize fields of Config (config.rs:67-70): serde takes any u64; a frame side beyond u32 is valid JSON this build cannot render
uint32_t as_frame_u32(const Value& v, const char* what) {
  const uint64_t x = as_u64(v, what, ~0ull);
  if (x > 0xFFFFFFFFull) throw UnsupportedError{std::string(what) + " = " + v.text + " does not fit 32 bits"};
  return uint32_t(x);
}

}  // namespace

// reference camera.rs:45-77
extern "C" void rt_camera_derive(const double lf[3], const double la[3], const double up[3], double vfov_deg,
                                 double aspect, double out[13]) {
  auto unit = [](const double v[3], double o[3]) {
    double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    o[0] = v[0] / l; o[1] = v[1] / l; o[2] = v[2] / l;
  };
  auto cross = [](const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
  };
  const double theta = vfov_deg * (3.14159265358979323846264338327950288 / 180.0);
  const double half_height = std::tan(theta / 2.0);
  const double half_width = aspect * half_height;
  double d[3] = {lf[0] - la[0], lf[1] - la[1], lf[2] - la[2]};
  double w[3], u[3], v[3], c[3];
  unit(d, w);
  cross(up, w, c);
  unit(c, u);
  cross(w, u, v);
  for (int i = 0; i < 3; ++i) {
    out[i] = lf[i];
    out[3 + i] = ((lf[i] - u[i] * half_width) - v[i] * half_height) - w[i];
    out[6 + i] = (u[i] * 2.0) * half_width;
    out[9 + i] = (v[i] * 2.0) * half_height;
  }
  out[12] = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}

// reference raytracer.rs:220-229
extern "C" uint32_t rt_find_lights(const RtSphere* spheres, uint32_t n, uint32_t* out_idx, uint32_t cap) {
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; ++i)
    if (spheres[i].kind == RT_MAT_LIGHT) {
      if (out_idx && k < cap) out_idx[k] = i;
      ++k;
    }
  return k;
}

struct RtSceneFile {
  RtScene scene{};
  std::vector<RtSphere> spheres;
  std::vector<RtTexture> textures;
  std::vector<std::unique_ptr<uint8_t, void (*)(void*)>> pixel_store;
  std::vector<std::string> texture_paths;  // per RtTexture
  std::unique_ptr<uint8_t, void (*)(void*)> sky_pixels{nullptr, std::free};
  std::string sky_path;
  double look_from[3]{}, look_at[3]{}, vup[3]{}, vfov = 0, aspect = 0, focal_length = 0;
  // where the load went (rt_scene_load_timings): reading the file, parsing the JSON text, the longest JPEG decode (they run
  // concurrently, beside the parse), everything (read + parse + schema + waiting for the decodes)
  double read_ms = 0, json_ms = 0, jpeg_ms = 0, total_ms = 0;
  std::shared_ptr<std::atomic<long long>> jpeg_us_max{new std::atomic<long long>(0)};
};

namespace {

// A decoded JPEG (or why it could not be decoded).  The scene's textures (2-3 MP each, ~40 ms of Huffman +
// IDCT) are decoded concurrently, one host thread per distinct file, while the rest of the scene is parsed.
struct Decoded {
  uint8_t* px = nullptr;
  uint32_t w = 0, h = 0;
  int rc = RT_OK;
  std::string err;
};
typedef std::map<std::string, std::shared_future<Decoded>> DecodeJobs;
thread_local std::shared_ptr<std::atomic<long long>> g_jpeg_us_max;  // the load in progress on this thread records its longest decode here
void start_decode(DecodeJobs& jobs, const std::string& path) {
  if (path.empty() || jobs.count(path)) return;
  std::shared_ptr<std::atomic<long long>> longest = g_jpeg_us_max;
  jobs[path] = std::async(std::launch::async, [path, longest]() {
    Decoded d;
    const auto t0 = std::chrono::steady_clock::now();
    d.rc = rt_jpeg_decode_file(path.c_str(), &d.px, &d.w, &d.h);
    if (d.rc != RT_OK) d.err = rt_jpeg_last_error();  // (thread-local in the decoder: read it on this thread)
    if (longest) {
      const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
      long long cur = longest->load();
      while (us > cur && !longest->compare_exchange_weak(cur, us)) {}
    }
    return d;
  }).share();
}
// every texture file the config names: Sky.texture and Texture.pixels of each object
void start_all_decodes(const Value& root, DecodeJobs& jobs) {
  if (root.kind != Value::Object) return;
  if (const Value* sky = root.find("sky"))
    if (sky->kind == Value::Object)
      if (const Value* t = sky->find("texture"))
        if (t->kind == Value::String) start_decode(jobs, t->text);
  const Value* objs = root.find("objects");
  if (!objs || objs->kind != Value::Array) return;
  for (auto& o : objs->items) {
    if (o->kind != Value::Object) continue;
    const Value* m = o->find("material");
    if (!m || m->kind != Value::Object || m->members.size() != 1 || m->members[0].first != "Texture") continue;
    const Value& body = *m->members[0].second;
    if (body.kind != Value::Object) continue;
    if (const Value* px = body.find("pixels"))
      if (px->kind == Value::String) start_decode(jobs, px->text);
  }
}
Decoded take_decoded(DecodeJobs& jobs, const std::string& path) {
  start_decode(jobs, path);
  return jobs[path].get();
}
// (a decode that nobody claimed — the config was rejected before its texture was reached — is freed here)
struct DecodeJobsGuard {
  DecodeJobs jobs;
  std::map<std::string, bool> taken;
  ~DecodeJobsGuard() {
    for (auto& j : jobs)
      if (!taken[j.first]) { Decoded d = j.second.get(); std::free(d.px); }
  }
};

// materials.rs:213-219 / config.rs:36-47; `File::open(path).expect(path)` -> RT_ERR_TEXTURE
uint32_t load_texture(RtSceneFile& sf, std::map<std::string, uint32_t>& cache, DecodeJobsGuard& dj, const std::string& path) {
  auto it = cache.find(path);
  if (it != cache.end()) return it->second;
  const Decoded d = take_decoded(dj.jobs, path);
  dj.taken[path] = true;
  if (d.rc != RT_OK) bad("texture " + path + ": " + d.err);
  uint8_t* px = d.px; const uint32_t w = d.w, h = d.h;
  sf.pixel_store.emplace_back(px, std::free);
  RtTexture t{};
  t.rgb8 = px; t.nbytes = uint64_t(w) * h * 3; t.width = w; t.height = h;
  sf.textures.push_back(t);
  sf.texture_paths.push_back(path);
  uint32_t id = uint32_t(sf.textures.size() - 1);
  cache[path] = id;
  return id;
}

void parse_albedo(const Value& v, float out[3]) {
  if (v.kind != Value::Array || v.items.size() != 3) bad("albedo: expected an array of length 3");
  for (int i = 0; i < 3; ++i) out[i] = as_f32(*v.items[i], "albedo");
}

void build_scene(const Value& root, RtSceneFile& sf) {
  const Value* cf[7];
  struct_fields(root, "Config", {"width", "height", "samples_per_pixel", "max_depth", "sky", "camera", "objects"}, cf,
                {false, false, false, false, true, false, false});
  RtScene& sc = sf.scene;
  sc.abi_version = RT_ABI_VERSION;
  sc.width = as_frame_u32(*cf[0], "width");
  sc.height = as_frame_u32(*cf[1], "height");
  sc.samples_per_pixel = uint32_t(as_u64(*cf[2], "samples_per_pixel", 0xFFFFFFFFull));   // u32 in the reference: beyond it serde errors too
  {
    const uint64_t d = as_u64(*cf[3], "max_depth", ~0ull);
    if (d > 0x7FFFFFFFull) throw UnsupportedError{"max_depth = " + cf[3]->text + " does not fit 31 bits"};
    sc.max_depth = uint32_t(d);
  }

  std::map<std::string, uint32_t> cache;
  DecodeJobsGuard dj;
  start_all_decodes(root, dj.jobs);
  // config.rs:22-28, 49-64: Option<Sky>; texture "" -> None
  const Value* sky = cf[4];
  sc.sky_mode = RT_SKY_NONE;
  if (sky && sky->kind != Value::Null) {
    const Value* sk[1];
    struct_fields(*sky, "Sky", {"texture"}, sk);
    const Value& tex = *sk[0];
    if (tex.kind != Value::String) bad("Sky.texture: expected a string");
    if (tex.text.empty()) sc.sky_mode = RT_SKY_GRADIENT;
    else {
      // (the sky gets its own copy of the pixels even when a sphere uses the same file: separate owners)
      Decoded d = take_decoded(dj.jobs, tex.text);
      if (d.rc != RT_OK) bad("sky texture " + tex.text + ": " + d.err);
      const uint32_t w = d.w, h = d.h;
      uint8_t* px = static_cast<uint8_t*>(std::malloc(size_t(w) * h * 3));
      if (!px) bad("sky texture " + tex.text + ": out of memory");
      std::memcpy(px, d.px, size_t(w) * h * 3);
      sf.sky_pixels.reset(px);
      sf.sky_path = tex.text;
      sc.sky_mode = RT_SKY_TEXTURE; sc.sky_rgb8 = px; sc.sky_w = w; sc.sky_h = h;
    }
  }

  // camera.rs:29-42 CameraParams -> Camera::new
  const Value* cam[5];
  struct_fields(*cf[5], "CameraParams", {"look_from", "look_at", "vup", "vfov", "aspect"}, cam);
  as_point(*cam[0], "camera.look_from", sf.look_from);
  as_point(*cam[1], "camera.look_at", sf.look_at);
  as_point(*cam[2], "camera.vup", sf.vup);
  sf.vfov = as_f64(*cam[3], "camera.vfov");
  sf.aspect = as_f64(*cam[4], "camera.aspect");
  double c[13];
  rt_camera_derive(sf.look_from, sf.look_at, sf.vup, sf.vfov, sf.aspect, c);
  std::memcpy(sc.cam_origin, c, 24); std::memcpy(sc.cam_lower_left, c + 3, 24);
  std::memcpy(sc.cam_horizontal, c + 6, 24); std::memcpy(sc.cam_vertical, c + 9, 24);
  sf.focal_length = c[12];

  const Value& objs = *cf[6];
  if (objs.kind != Value::Array) bad("objects: expected an array");
  if (objs.items.size() > 0xFFFFFFFFull) throw UnsupportedError{"more than 2^32 - 1 objects"};
  sf.spheres.reserve(objs.items.size());
  for (auto& o : objs.items) {
    RtSphere s{};
    const Value* sp[3];
    struct_fields(*o, "Sphere", {"center", "radius", "material"}, sp);
    as_point(*sp[0], "Sphere.center", s.center);
    s.radius = as_f64(*sp[1], "Sphere.radius");
    const Value& m = *sp[2];
    // externally tagged enum (materials.rs:35-42): a map with exactly one key (a second one — also the same one twice — is an error)
    if (m.kind != Value::Object || m.members.size() != 1) bad("material: expected a single-key enum object");
    const std::string& tag = m.members[0].first;
    const Value& body = *m.members[0].second;
    const Value* f[5];
    if (tag == "Lambertian") {
      struct_fields(body, "Lambertian", {"albedo"}, f);
      s.kind = RT_MAT_LAMBERTIAN; parse_albedo(*f[0], s.albedo);
    } else if (tag == "Metal") {
      struct_fields(body, "Metal", {"albedo", "fuzz"}, f);
      s.kind = RT_MAT_METAL; parse_albedo(*f[0], s.albedo);
      s.fuzz_or_ior = as_f64(*f[1], "Metal.fuzz");
    } else if (tag == "Glass") {
      struct_fields(body, "Glass", {"index_of_refraction"}, f);
      s.kind = RT_MAT_GLASS;
      s.fuzz_or_ior = as_f64(*f[0], "Glass.index_of_refraction");
    } else if (tag == "Texture") {
      struct_fields(body, "Texture", {"albedo", "pixels", "width", "height", "h_offset"}, f);
      s.kind = RT_MAT_TEXTURE; parse_albedo(*f[0], s.albedo);
      if (f[1]->kind != Value::String) bad("Texture.pixels: expected a path string");
      s.tex_w = as_u64(*f[2], "Texture.width", ~0ull);
      s.tex_h = as_u64(*f[3], "Texture.height", ~0ull);
      s.h_offset = as_f64(*f[4], "Texture.h_offset");
      s.tex_id = load_texture(sf, cache, dj, f[1]->text);
    } else if (tag == "Light") {
      struct_fields(body, "Light", {}, f);   // materials.rs:56-57: `struct Light {}` — {} or []
      s.kind = RT_MAT_LIGHT;
    } else {
      bad("unknown variant `" + tag + "`, expected one of `Lambertian`, `Metal`, `Glass`, `Texture`, `Light`");
    }
    sf.spheres.push_back(s);
  }
  sc.spheres = sf.spheres.data(); sc.n_spheres = uint32_t(sf.spheres.size());
  sc.textures = sf.textures.data(); sc.n_textures = uint32_t(sf.textures.size());
  sc.seed = 0;
}

// ---- serde_json::to_string number formatting (ryu "pretty" layout) ----
template <typename F>
std::string shortest(F x, int max_prec, bool is32) {
  if (x == 0) return std::signbit(x) ? "-0.0" : "0.0";
  if (!std::isfinite(x)) return "null";  // serde_json writes non-finite floats as null (an f32 field cast from 1e39 is +inf)
  char buf[64];
  int prec = 0;
  for (; prec <= max_prec; ++prec) {
    std::snprintf(buf, sizeof buf, "%.*e", prec, double(x));
    if (is32 ? (std::strtof(buf, nullptr) == float(x)) : (std::strtod(buf, nullptr) == double(x))) break;
  }
  // buf = d.ddddde[+-]XX
  std::string s(buf);
  bool neg = s[0] == '-';
  if (neg) s.erase(0, 1);
  size_t epos = s.find('e');
  int exp10 = std::atoi(s.c_str() + epos + 1);
  std::string digits;
  for (size_t i = 0; i < epos; ++i) if (s[i] != '.') digits += s[i];
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  const int len = int(digits.size());
  const int kk = exp10 + 1;      // position of the decimal point relative to digits start
  const int k = kk - len;        // exponent of the last digit
  const int hi = is32 ? 13 : 16, lo = is32 ? -6 : -5;
  std::string out = neg ? "-" : "";
  if (0 <= k && kk <= hi) { out += digits + std::string(k, '0') + ".0"; }
  else if (0 < kk && kk <= hi) { out += digits.substr(0, kk) + "." + digits.substr(kk); }
  else if (lo < kk && kk <= 0) { out += "0." + std::string(-kk, '0') + digits; }
  else if (len == 1) { out += digits + "e" + std::to_string(kk - 1); }
  else { out += digits.substr(0, 1) + "." + digits.substr(1) + "e" + std::to_string(kk - 1); }
  return out;
}
std::string f64s(double x) { return shortest<double>(x, 17, false); }
std::string f32s(float x) { return shortest<float>(x, 9, true); }
std::string jstr(const std::string& s) {
  std::string o = "\"";
  for (unsigned char c : s) {
    if (c == '"') o += "\\\"";
    else if (c == '\\') o += "\\\\";
    else if (c == '\n') o += "\\n";
    else if (c == '\r') o += "\\r";
    else if (c == '\t') o += "\\t";
    else if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); o += b; }
    else o += char(c);
  }
  return o + "\"";
}
std::string point(const double p[3]) {
  return "{\"x\":" + f64s(p[0]) + ",\"y\":" + f64s(p[1]) + ",\"z\":" + f64s(p[2]) + "}";
}
std::string albedo(const float a[3]) { return "[" + f32s(a[0]) + "," + f32s(a[1]) + "," + f32s(a[2]) + "]"; }

std::string scene_json(const RtSceneFile& sf) {
  const RtScene& sc = sf.scene;
  std::string o = "{\"width\":" + std::to_string(sc.width) + ",\"height\":" + std::to_string(sc.height) +
                  ",\"samples_per_pixel\":" + std::to_string(sc.samples_per_pixel) +
                  ",\"max_depth\":" + std::to_string(sc.max_depth) + ",\"sky\":";
  if (sc.sky_mode == RT_SKY_NONE) o += "null";
  else o += "{\"texture\":" + jstr(sc.sky_mode == RT_SKY_TEXTURE ? sf.sky_path : std::string()) + "}";
  o += ",\"camera\":{\"look_from\":" + point(sf.look_from) + ",\"look_at\":" + point(sf.look_at) +
       ",\"vup\":" + point(sf.vup) + ",\"vfov\":" + f64s(sf.vfov) + ",\"aspect\":" + f64s(sf.aspect) + "},\"objects\":[";
  for (size_t i = 0; i < sf.spheres.size(); ++i) {
    const RtSphere& s = sf.spheres[i];
    if (i) o += ",";
    o += "{\"center\":" + point(s.center) + ",\"radius\":" + f64s(s.radius) + ",\"material\":{";
    switch (s.kind) {
      case RT_MAT_LAMBERTIAN: o += "\"Lambertian\":{\"albedo\":" + albedo(s.albedo) + "}"; break;
      case RT_MAT_METAL: o += "\"Metal\":{\"albedo\":" + albedo(s.albedo) + ",\"fuzz\":" + f64s(s.fuzz_or_ior) + "}"; break;
      case RT_MAT_GLASS: o += "\"Glass\":{\"index_of_refraction\":" + f64s(s.fuzz_or_ior) + "}"; break;
      case RT_MAT_TEXTURE:  // materials.rs:28-33: pixels always serialize as "/tmp/texture.jpg"
        o += "\"Texture\":{\"albedo\":" + albedo(s.albedo) + ",\"pixels\":\"/tmp/texture.jpg\",\"width\":" +
             std::to_string(s.tex_w) + ",\"height\":" + std::to_string(s.tex_h) + ",\"h_offset\":" + f64s(s.h_offset) + "}";
        break;
      default: o += "\"Light\":{}"; break;
    }
    o += "}}";
  }
  return o + "]}";
}

}  // namespace

extern "C" const char* rt_host_last_error(void) { return g_err.c_str(); }

extern "C" int rt_scene_load_string(const char* json_text, size_t len, RtSceneFile** out) {
  if (!json_text || !out) return set_err(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  std::unique_ptr<RtSceneFile> sf(new RtSceneFile);
  const auto t0 = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  g_jpeg_us_max = sf->jpeg_us_max;
  struct Reset { ~Reset() { g_jpeg_us_max.reset(); } } reset;
  try {
    rtjson::ValuePtr root = rtjson::parse(json_text, len);
    sf->json_ms = ms_since(t0);
    build_scene(*root, *sf);
  } catch (const rtjson::ParseError& e) {
    return set_err(RT_ERR_PARSE, std::string("Unable to parse config json: ") + e.what());
  } catch (const UnsupportedError& e) {
    return set_err(RT_ERR_UNSUPPORTED, "config is valid but unsupported: " + e.msg);
  } catch (const SchemaError& e) {
    bool tex = e.msg.rfind("texture ", 0) == 0 || e.msg.rfind("sky texture ", 0) == 0;
    return set_err(tex ? RT_ERR_TEXTURE : RT_ERR_PARSE, (tex ? std::string() : std::string("Unable to parse config json: ")) + e.msg);
  }
  sf->jpeg_ms = (double)sf->jpeg_us_max->load() * 1e-3;
  sf->total_ms = ms_since(t0);
  *out = sf.release();
  return RT_OK;
}

extern "C" void rt_scene_load_timings(const RtSceneFile* sf, double out[4]) {
  if (!sf || !out) return;
  out[0] = sf->read_ms; out[1] = sf->json_ms; out[2] = sf->jpeg_ms; out[3] = sf->total_ms;
}

extern "C" int rt_scene_load_file(const char* json_path, RtSceneFile** out) {
  if (!json_path || !out) return set_err(RT_ERR_INVALID, "null argument");
  FILE* f = std::fopen(json_path, "rb");
  if (!f) return set_err(RT_ERR_IO, std::string("Unable to read config file. (") + json_path + ": " + std::strerror(errno) + ")");
  const auto t0 = std::chrono::steady_clock::now();
  std::string text; char buf[65536]; size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
  std::fclose(f);
  const double read_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  const int rc = rt_scene_load_string(text.data(), text.size(), out);
  if (rc == RT_OK) { (*out)->read_ms = read_ms; (*out)->total_ms += read_ms; }
  return rc;
}

extern "C" const RtScene* rt_scene_get(const RtSceneFile* sf) { return sf ? &sf->scene : nullptr; }
extern "C" RtScene* rt_scene_get_mut(RtSceneFile* sf) { return sf ? &sf->scene : nullptr; }
extern "C" void rt_scene_free(RtSceneFile* sf) { delete sf; }
extern "C" void rt_scene_camera(const RtSceneFile* sf, double out[11]) {
  if (!sf || !out) return;
  for (int i = 0; i < 3; ++i) { out[i] = sf->look_from[i]; out[3 + i] = sf->look_at[i]; out[6 + i] = sf->vup[i]; }
  out[9] = sf->vfov; out[10] = sf->aspect;
}
extern "C" void rt_free(void* p) { std::free(p); }

extern "C" int rt_scene_to_json(const RtSceneFile* sf, char* buf, size_t cap, size_t* needed) {
  if (!sf) return set_err(RT_ERR_INVALID, "null scene");
  std::string s = scene_json(*sf);
  if (needed) *needed = s.size() + 1;
  if (buf && cap) {
    size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return (buf && cap > s.size()) || !buf ? RT_OK : RT_ERR_INVALID;
}

// reference raytracer.rs:33-42 write_image: PNG, 8-bit RGB, non-interlaced.  The reference's encoder output (image 0.13 /
// png crate) is not a byte contract — the decoded pixels are; this writer produces them from any thread count.
//
// The renderer needs 12.7 ms for the headline frame; a single-threaded zlib level-6 pass over it takes ~135 ms (1.2 s at
// 4K), and the banded level-6 writer of rounds 1 - 5 still 20 ms on the GPU box — LONGER than the kernel, so the frame rate
// of an animation (README.md:43-57) was the PNG writer's.  Round 6:
//   * a path-traced frame is noise on gradients: after the Sub filter nearly all of deflate's gain is the Huffman coding
//     of small residuals plus runs of equal bytes; LZ77 matching beyond distance 1 buys 2.6 % of file size for 4.7 x the time
//     (measured on the headline frame: level 6 1.260 MB / 220 ms of one core, Z_RLE 1.293 MB / 47 ms).  Default strategy:
//     Z_RLE (RT_PNG_DEFLATE=default gives the old level-6 files, =huffman Huffman only);
//   * the scanlines are filtered and deflated in independent bands of ~48 KB on the host's cores (raw deflate, each band
//     closed by a sync flush so that it ends on a byte boundary, the last one finishes the stream) — with distance-1 matches
//     a band loses nothing by starting with an empty window — and every band is its OWN IDAT chunk, CRC computed by the
//     thread that made it (consecutive IDAT chunks are one zlib stream: PNG spec 11.2.4); the stream's Adler-32 is combined
//     from the bands' and travels in a last 4-byte IDAT chunk;
//   * one z_stream and one scratch buffer per thread (deflateReset between bands), the filter loop vectorised.
namespace {
struct PngBand {
  std::vector<uint8_t> chunk;  // complete IDAT chunk: length, "IDAT", deflate bytes, CRC
  uLong adler = 1;
  size_t raw_len = 0;
  bool ok = false;
};
void put_be32(uint8_t* p, uint32_t v) { p[0] = uint8_t(v >> 24); p[1] = uint8_t(v >> 16); p[2] = uint8_t(v >> 8); p[3] = uint8_t(v); }

// filter type 1 (Sub): residual against the pixel to the left — compresses rendered images well, and (unlike Up / Paeth) a band
// needs no scanline of its neighbour
__attribute__((optimize("tree-vectorize"))) void png_filter_sub(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t stride) {
  dst[0] = 1;
  const size_t head = stride < 3 ? stride : 3;
  for (size_t i = 0; i < head; ++i) dst[1 + i] = src[i];
  for (size_t i = 3; i < stride; ++i) dst[1 + i] = uint8_t(src[i] - src[i - 3]);
}

struct PngWorker {   // per thread: a deflate state and the band's filtered bytes, reused from band to band
  z_stream zs;
  bool live = false;
  std::vector<uint8_t> raw;
  ~PngWorker() { if (live) deflateEnd(&zs); }
  bool begin(int level, int strategy) {
    if (live) return deflateReset(&zs) == Z_OK;
    std::memset(&zs, 0, sizeof zs);
    // -15: raw deflate, no zlib wrapper (the wrapper is written once for the whole stream)
    live = deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) == Z_OK;
    return live;
  }
};
void png_deflate_band(PngWorker& wk, int level, int strategy, const uint8_t* rgb8, uint32_t w, uint32_t y0, uint32_t y1, bool first, bool last, PngBand& out) {
  const size_t stride = size_t(w) * 3;
  wk.raw.resize((stride + 1) * (y1 - y0));
  for (uint32_t y = y0; y < y1; ++y) png_filter_sub(rgb8 + stride * y, &wk.raw[(stride + 1) * (y - y0)], stride);
  out.raw_len = wk.raw.size();
  out.adler = adler32(adler32(0L, Z_NULL, 0), wk.raw.data(), uInt(wk.raw.size()));
  if (!wk.begin(level, strategy)) return;
  const size_t head = 8 + (first ? 2 : 0);  // chunk length + type (+ the zlib header in the first band)
  out.chunk.resize(head + deflateBound(&wk.zs, uLong(wk.raw.size())) + 16 + 4);
  wk.zs.next_in = wk.raw.data(); wk.zs.avail_in = uInt(wk.raw.size());
  wk.zs.next_out = out.chunk.data() + head; wk.zs.avail_out = uInt(out.chunk.size() - head - 4);
  const int rc = deflate(&wk.zs, last ? Z_FINISH : Z_SYNC_FLUSH);
  out.ok = (last ? rc == Z_STREAM_END : rc == Z_OK) && wk.zs.avail_in == 0;
  const size_t zlen = out.chunk.size() - head - 4 - wk.zs.avail_out, data_len = zlen + (first ? 2 : 0);
  out.chunk.resize(8 + data_len + 4);
  put_be32(out.chunk.data(), uint32_t(data_len));
  std::memcpy(out.chunk.data() + 4, "IDAT", 4);
  if (first) { out.chunk[8] = 0x78; out.chunk[9] = 0x9C; }  // zlib header: deflate, 32 KB window, default level, no dictionary
  put_be32(out.chunk.data() + 8 + data_len, uint32_t(crc32(0L, out.chunk.data() + 4, uInt(4 + data_len))));
}
}  // namespace

extern "C" int rt_png_write_rgb8(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h) {
  if (!path || !rgb8 || !w || !h) return set_err(RT_ERR_INVALID, "bad png arguments");
  const size_t stride = size_t(w) * 3;
  if ((stride + 1) * size_t(h) > 0x7FFFFFFFu) return set_err(RT_ERR_PNG, "image too large");
  int level = 1, strategy = Z_RLE;
  if (const char* e = std::getenv("RT_PNG_DEFLATE")) {
    if (!std::strcmp(e, "default")) { level = 6; strategy = Z_DEFAULT_STRATEGY; }
    else if (!std::strcmp(e, "huffman")) strategy = Z_HUFFMAN_ONLY;
    else if (std::strcmp(e, "rle")) return set_err(RT_ERR_INVALID, "RT_PNG_DEFLATE must be rle (default), default or huffman");
  }
  // threads: the host's cores, at most RT_PNG_THREADS (default 32, 64 for frames beyond 8 MB: beyond that starting the threads costs what
  // they save — headline frame 1.85 ms on 16 or 32 threads, 1.93 on 64; a 4K frame 7.1 ms on 32, 5.0 on 64, 5.8 on 128; tools/png_bench.py)
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 4;
  unsigned cap = (stride + 1) * size_t(h) > (size_t(8) << 20) ? 64 : 32;
  if (const char* e = std::getenv("RT_PNG_THREADS")) { const long v = std::strtol(e, nullptr, 10); if (v >= 1 && v <= 1024) cap = unsigned(v); }
  hw = std::min(hw, cap);
  // bands of >= 48 KB of scanlines (128 KB with LZ77 matching: a band starts with an empty window), at most 8 per thread
  const size_t band_bytes = strategy == Z_DEFAULT_STRATEGY ? size_t(128) * 1024 : size_t(48) * 1024;
  uint32_t rows_per_band = uint32_t((band_bytes + stride) / (stride + 1));
  if (rows_per_band == 0) rows_per_band = 1;
  uint32_t n_bands = (h + rows_per_band - 1) / rows_per_band;
  if (n_bands > 8 * hw) { n_bands = 8 * hw; rows_per_band = (h + n_bands - 1) / n_bands; n_bands = (h + rows_per_band - 1) / rows_per_band; }
  std::vector<PngBand> bands(n_bands);
  {
    std::atomic<uint32_t> next{0};
    auto work = [&]() {
      PngWorker wk;
      for (uint32_t b; (b = next.fetch_add(1)) < n_bands;) {
        const uint32_t y0 = b * rows_per_band, y1 = std::min(h, y0 + rows_per_band);
        png_deflate_band(wk, level, strategy, rgb8, w, y0, y1, b == 0, b + 1 == n_bands, bands[b]);
      }
    };
    const unsigned n_threads = std::min<unsigned>(hw, n_bands);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
  }
  size_t total = 8 + 25 + 16 + 12;  // signature, IHDR, the Adler-32's IDAT, IEND
  uLong adler = adler32(0L, Z_NULL, 0);
  for (const PngBand& b : bands) {
    if (!b.ok) return set_err(RT_ERR_PNG, "error writing image (deflate)");
    total += b.chunk.size();
    adler = adler32_combine(adler, b.adler, z_off_t(b.raw_len));
  }
  std::vector<uint8_t> file(total);
  uint8_t* p = file.data();
  auto chunk = [&](const char* type, const uint8_t* data, uint32_t n) {
    put_be32(p, n); std::memcpy(p + 4, type, 4);
    if (n) std::memcpy(p + 8, data, n);
    put_be32(p + 8 + n, uint32_t(crc32(0L, p + 4, 4 + n)));
    p += 12 + n;
  };
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
  std::memcpy(p, sig, 8); p += 8;
  uint8_t ihdr[13]; put_be32(ihdr, w); put_be32(ihdr + 4, h);
  ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk("IHDR", ihdr, 13);
  for (const PngBand& b : bands) { std::memcpy(p, b.chunk.data(), b.chunk.size()); p += b.chunk.size(); }
  uint8_t tail[4]; put_be32(tail, uint32_t(adler));
  chunk("IDAT", tail, 4);
  chunk("IEND", nullptr, 0);
  FILE* f = std::fopen(path, "wb");
  if (!f) return set_err(RT_ERR_PNG, std::string("error writing image: ") + std::strerror(errno));
  bool ok = std::fwrite(file.data(), 1, total, f) == total;
  ok = (std::fclose(f) == 0) && ok;
  return ok ? RT_OK : set_err(RT_ERR_PNG, "error writing image");
}
