// Library-level entry points: ABI version, error text, capacity arithmetic.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace eve {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace eve

extern "C" int eve_moe_abi_version(void) { return EVE_MOE_ABI_VERSION; }

extern "C" const char* eve_moe_last_error(void) { return eve::g_err; }

#ifndef EVE_SOURCE_HASH
#define EVE_SOURCE_HASH "unknown"
#endif
extern "C" const char* eve_moe_source_hash(void) { return EVE_SOURCE_HASH; }

// models/sharded_moe.py:150-160 as called at :177: the python float S/E is multiplied by a
// 0-dim fp32 tensor holding 2*capacity_factor, so product and ceil are fp32 arithmetic.
extern "C" int eve_capacity(int S, int E, double capacity_factor, int min_capacity) {
  if (S <= 0 || E <= 0) return min_capacity;
  const float per_expert = (float)((double)S / (double)E);
  const float factor = (float)(capacity_factor * 2.0);
  const float prod = per_expert * factor;
  long long c = (long long)ceilf(prod);
  if (c < (long long)min_capacity) c = min_capacity;
  if (c > 0x7fffffffLL) c = 0x7fffffffLL;
  return (int)c;
}
