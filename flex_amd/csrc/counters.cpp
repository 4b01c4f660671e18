// libflex_counters.so: card-wide hardware counters read by the process itself (include/flex_counters.h).
//
// The rocprofiler-sdk device counting service: one context per GPU, configured while the profiler initialises
// (tool_init below; the profiler comes up with the first ROCm runtime of the process and asks rocprofiler_configure
// for it -- rocprofiler_force_configure ahead of the runtime aborts inside hsa_init on ROCm 7.2), each with a callback that hands the service the
// counter set of the pass being started.  begin = choose the set + start the context (the service programs and
// starts the counters), end = sample (reads what accumulated since the start) + stop.  Nothing is attached to
// the dispatches, so the launches between begin and end run as they do without the profiler.
#include "../../include/flex_counters.h"

#include <rocprofiler-sdk/registration.h>
#include <rocprofiler-sdk/rocprofiler.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string t_error;

struct Card {
  rocprofiler_agent_id_t agent{};
  uint32_t node = 0;  // logical order of rocminfo
  rocprofiler_context_id_t ctx{};
  rocprofiler_buffer_id_t buf{};
  rocprofiler_counter_config_id_t config{};  // the set of the pass being started; handle 0 = none
  std::map<std::string, rocprofiler_counter_id_t> by_name;
  std::map<uint64_t, size_t> instances;  // counter id -> records one sample of it returns
};

std::mutex g_lock;
std::vector<Card> g_cards;  // filled once by tool_init, never resized afterwards (the service keeps pointers into it)
std::atomic<int> g_state{0};  // 0 not initialised, 1 ready (tool_init ran), <0 the error init returns from then on
bool g_wanted = false;      // flex_counters_init was called in time: rocprofiler_configure hands the profiler our tool
std::string g_init_error;
int g_open = -1;                      // card of the pass in flight
std::vector<uint64_t> g_open_ids;     // its counters, in the caller's order
size_t g_open_records = 0;

bool ok(rocprofiler_status_t s, const char *what, std::string &err) {
  if (s == ROCPROFILER_STATUS_SUCCESS) return true;
  err = std::string(what) + ": " + rocprofiler_get_status_string(s);
  return false;
}

bool list_counters(Card &c, std::string &err) {
  std::vector<rocprofiler_counter_id_t> ids;
  if (!ok(rocprofiler_iterate_agent_supported_counters(
              c.agent,
              [](rocprofiler_agent_id_t, rocprofiler_counter_id_t *counters, size_t n, void *user) {
                auto *v = static_cast<std::vector<rocprofiler_counter_id_t> *>(user);
                v->insert(v->end(), counters, counters + n);
                return ROCPROFILER_STATUS_SUCCESS;
              },
              &ids),
          "listing the card's counters", err))
    return false;
  for (auto id : ids) {
    rocprofiler_counter_info_v1_t info;
    if (!ok(rocprofiler_query_counter_info(id, ROCPROFILER_COUNTER_INFO_VERSION_1, &info), "counter info", err)) return false;
    c.by_name.emplace(info.name, id);
    c.instances.emplace(id.handle, info.dimensions_instances_count);
  }
  return true;
}

int tool_init_body();
int tool_init(rocprofiler_client_finalize_t, void *) {
  int rc = tool_init_body();  // called by the profiler while the runtime initialises (g_lock is not held here)
  g_state = rc == 0 && !g_cards.empty() ? 1 : FLEX_COUNTERS_ERR_PROFILER;
  if (g_state != 1 && g_init_error.empty()) g_init_error = "the profiler lists no GPU";
  return rc;
}
int tool_init_body() {
  std::string &err = g_init_error;
  std::vector<Card> cards;
  rocprofiler_query_available_agents_cb_t each = [](rocprofiler_agent_version_t, const void **agents, size_t n, void *user) {
    auto *out = static_cast<std::vector<Card> *>(user);
    for (size_t i = 0; i < n; ++i) {
      const auto *a = static_cast<const rocprofiler_agent_v0_t *>(agents[i]);
      if (a->type != ROCPROFILER_AGENT_TYPE_GPU) continue;
      Card c;
      c.agent = a->id;
      c.node = a->logical_node_type_id;
      out->push_back(c);
    }
    return ROCPROFILER_STATUS_SUCCESS;
  };
  if (!ok(rocprofiler_query_available_agents(ROCPROFILER_AGENT_INFO_VERSION_0, each, sizeof(rocprofiler_agent_t), &cards),
          "listing the agents", err))
    return -1;
  std::sort(cards.begin(), cards.end(), [](const Card &a, const Card &b) { return a.node < b.node; });
  g_cards = std::move(cards);
  for (Card &c : g_cards) {
    rocprofiler_callback_thread_t thread{};
    if (!ok(rocprofiler_create_context(&c.ctx), "creating a context", err)) return -1;
    // the service wants a buffer even though a synchronous sample returns its records directly
    if (!ok(rocprofiler_create_buffer(
                c.ctx, 4096, 2048, ROCPROFILER_BUFFER_POLICY_LOSSLESS,
                [](rocprofiler_context_id_t, rocprofiler_buffer_id_t, rocprofiler_record_header_t **, size_t, void *, uint64_t) {},
                nullptr, &c.buf),
            "creating a buffer", err) ||
        !ok(rocprofiler_create_callback_thread(&thread), "creating the callback thread", err) ||
        !ok(rocprofiler_assign_callback_thread(c.buf, thread), "assigning the callback thread", err))
      return -1;
    if (!ok(rocprofiler_configure_device_counting_service(
                c.ctx, c.buf, c.agent,
                [](rocprofiler_context_id_t ctx, rocprofiler_agent_id_t, rocprofiler_device_counting_agent_cb_t set, void *user) {
                  auto *card = static_cast<Card *>(user);
                  if (card->config.handle != 0) set(ctx, card->config);
                },
                &c),
            "configuring the device counting service", err))
      return -1;
  }
  return 0;
}

}  // namespace

// Found by rocprofiler-register (dlsym over the process) when the first ROCm runtime initialises: the profiler is
// brought up only if flex_counters_init asked for it before that moment.
extern "C" __attribute__((visibility("default"))) rocprofiler_tool_configure_result_t *rocprofiler_configure(
    uint32_t, const char *, uint32_t, rocprofiler_client_id_t *id) {
  std::lock_guard<std::mutex> hold(g_lock);
  if (!g_wanted) return nullptr;
  id->name = "flex_counters";
  static rocprofiler_tool_configure_result_t cfg{sizeof(rocprofiler_tool_configure_result_t), &tool_init, nullptr, nullptr};
  return &cfg;
}

extern "C" {

int flex_counters_init(void) {
  std::lock_guard<std::mutex> hold(g_lock);
  if (g_state == 1 || (g_state == 0 && g_wanted)) return FLEX_COUNTERS_OK;
  if (g_state < 0) {
    t_error = g_init_error;
    return g_state;
  }
  int up = 0;
  rocprofiler_is_initialized(&up);
  if (up != 0) {  // a runtime came up before anybody asked: the profiler was not, and cannot be, attached any more
    g_init_error = "a ROCm runtime initialised before flex_counters_init (call it before the first HIP call of the process)";
    t_error = g_init_error;
    return g_state = FLEX_COUNTERS_ERR_LATE;
  }
  g_wanted = true;  // the contexts are created by tool_init, when the runtime initialises
  return FLEX_COUNTERS_OK;
}

int flex_counters_devices(void) {
  std::lock_guard<std::mutex> hold(g_lock);
  return g_state == 1 ? (int)g_cards.size() : 0;
}

int flex_counters_begin(int device, const char *const *names, int n) {
  std::lock_guard<std::mutex> hold(g_lock);
  if (g_state != 1 || g_open >= 0 || device < 0 || device >= (int)g_cards.size() || !names || n <= 0) {
    t_error = g_state < 0 ? g_init_error : g_state != 1 ? "the profiler is not up: flex_counters_init, then one HIP call (the runtime brings it up), then begin" : g_open >= 0 ? "a pass is already open" : "bad arguments";
    return FLEX_COUNTERS_ERR_STATE;
  }
  Card &c = g_cards[device];
  if (c.by_name.empty() && !list_counters(c, t_error)) return FLEX_COUNTERS_ERR_PROFILER;
  std::vector<rocprofiler_counter_id_t> ids;
  size_t records = 0;
  for (int i = 0; i < n; ++i) {
    auto it = names[i] ? c.by_name.find(names[i]) : c.by_name.end();
    if (it == c.by_name.end()) {
      t_error = std::string("no counter named ") + (names[i] ? names[i] : "(null)");
      return FLEX_COUNTERS_ERR_NAME;
    }
    ids.push_back(it->second);
    records += c.instances[it->second.handle];
  }
  rocprofiler_counter_config_id_t config{};
  if (!ok(rocprofiler_create_counter_config(c.agent, ids.data(), ids.size(), &config), "creating the counter set (too many for one pass?)",
          t_error))
    return FLEX_COUNTERS_ERR_PROFILER;
  c.config = config;
  if (!ok(rocprofiler_start_context(c.ctx), "starting the counters", t_error)) {
    c.config.handle = 0;
    return FLEX_COUNTERS_ERR_PROFILER;
  }
  g_open = device;
  g_open_ids.clear();
  for (auto id : ids) g_open_ids.push_back(id.handle);
  g_open_records = records;
  return FLEX_COUNTERS_OK;
}

int flex_counters_end(double *values) {
  std::lock_guard<std::mutex> hold(g_lock);
  if (g_open < 0 || !values) {
    t_error = g_open < 0 ? "no pass is open" : "bad arguments";
    return FLEX_COUNTERS_ERR_STATE;
  }
  Card &c = g_cards[g_open];
  std::vector<rocprofiler_counter_record_t> rec(g_open_records + 64);
  size_t got = rec.size();
  rocprofiler_status_t s = rocprofiler_sample_device_counting_service(c.ctx, {}, ROCPROFILER_COUNTER_FLAG_NONE, rec.data(), &got);
  rocprofiler_stop_context(c.ctx);
  c.config.handle = 0;
  g_open = -1;
  if (!ok(s, "reading the counters", t_error)) return FLEX_COUNTERS_ERR_PROFILER;
  for (size_t i = 0; i < g_open_ids.size(); ++i) values[i] = 0.0;
  for (size_t r = 0; r < got; ++r) {
    rocprofiler_counter_id_t id{};
    if (rocprofiler_query_record_counter_id(rec[r].id, &id) != ROCPROFILER_STATUS_SUCCESS) continue;
    for (size_t i = 0; i < g_open_ids.size(); ++i)
      if (g_open_ids[i] == id.handle) values[i] += rec[r].counter_value;
  }
  return FLEX_COUNTERS_OK;
}

const char *flex_counters_error(void) { return t_error.c_str(); }

}  // extern "C"
