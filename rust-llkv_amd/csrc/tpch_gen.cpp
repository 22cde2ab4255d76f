// Deterministic TPC-H-shaped generator (harness side; see include/llkv_tpch_gen.h).
//
// Shapes follow SURVEY.md §8(d): 1–7 lines per order in a fixed 7-order / 28-line
// period (so row → order is closed-form), dbgen-style sparse order keys, retail
// price formula of dbgen, prices/discounts/taxes built as (double)integer / 100.0
// so they compare equal to the literal casts of llkv-types/src/literal.rs:487-492.
#include "llkv_tpch_gen.h"

#include <algorithm>
#include <thread>
#include <vector>

namespace {

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

constexpr uint64_t STREAM_ORDER = 1ULL << 60;
constexpr uint64_t STREAM_LINE_A = 2ULL << 60;
constexpr uint64_t STREAM_LINE_B = 3ULL << 60;
constexpr uint64_t STREAM_CUSTOMER = 4ULL << 60;

// Orders of one period have 1,2,...,7 lines: cumulative starts inside the 28-line period.
constexpr uint32_t PERIOD_LINES = 28;
constexpr uint32_t PERIOD_ORDERS = 7;
constexpr uint8_t ORDER_OF_LINE[PERIOD_LINES] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4,
                                                 4, 5, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 6, 6};
constexpr uint8_t FIRST_LINE_OF_ORDER[PERIOD_ORDERS] = {0, 1, 3, 6, 10, 15, 21};

inline uint64_t order_index_of_row(uint64_t row) {
  return (row / PERIOD_LINES) * PERIOD_ORDERS + ORDER_OF_LINE[row % PERIOD_LINES];
}

inline int64_t order_key_of_index(uint64_t j) { return (int64_t)((j / 8) * 32 + (j % 8) + 1); }

inline int32_t order_date_of_index(uint64_t seed, uint64_t j) {
  uint64_t h = splitmix64(STREAM_ORDER ^ seed ^ j);
  return LLKV_TPCH_DATE_1992_01_01 + (int32_t)(h % 2406); // .. 1998-08-02
}

inline uint64_t scaled(double scale, double per_sf) {
  double v = per_sf * scale;
  return v < 1.0 ? 1ULL : (uint64_t)v;
}

template <class F> void parallel_rows(uint64_t rows, int32_t threads, F &&body) {
  uint32_t t = threads > 0 ? (uint32_t)threads : std::max(1u, std::thread::hardware_concurrency());
  if (rows < (1u << 16)) t = 1;
  if (t == 1) {
    body(0, rows);
    return;
  }
  std::vector<std::thread> pool;
  uint64_t per = (rows + t - 1) / t;
  for (uint32_t k = 0; k < t; ++k) {
    uint64_t b = std::min(rows, k * per), e = std::min(rows, b + per);
    if (b < e) pool.emplace_back([=, &body] { body(b, e); });
  }
  for (auto &th : pool) th.join();
}

} // namespace

extern "C" {

uint64_t llkv_tpch_orders_for_lineitems(uint64_t lineitem_rows) {
  return lineitem_rows == 0 ? 0 : order_index_of_row(lineitem_rows - 1) + 1;
}

uint64_t llkv_tpch_customers_for_scale(double scale) { return scaled(scale, 150000.0); }

void llkv_tpch_gen_lineitem(uint64_t seed, double scale, uint64_t row_begin, uint64_t rows,
                            int64_t *l_orderkey, int64_t *l_partkey, int64_t *l_suppkey,
                            int64_t *l_linenumber, int64_t *l_quantity, double *l_extendedprice,
                            double *l_discount, double *l_tax, int32_t *l_shipdate,
                            int32_t *l_commitdate, int32_t *l_receiptdate, uint8_t *l_returnflag,
                            uint8_t *l_linestatus, int32_t threads) {
  const uint64_t n_part = scaled(scale, 200000.0);
  const uint64_t n_supp = scaled(scale, 10000.0);
  parallel_rows(rows, threads, [&](uint64_t b, uint64_t e) {
    for (uint64_t k = b; k < e; ++k) {
      const uint64_t i = row_begin + k;
      const uint32_t o = (uint32_t)(i % PERIOD_LINES);
      const uint32_t ord = ORDER_OF_LINE[o];
      const uint64_t j = (i / PERIOD_LINES) * PERIOD_ORDERS + ord;
      const uint64_t ha = splitmix64(STREAM_LINE_A ^ seed ^ i);
      const uint64_t hb = splitmix64(STREAM_LINE_B ^ seed ^ i);
      const int32_t odate = order_date_of_index(seed, j);
      const int64_t qty = 1 + (int64_t)(ha % 50);
      const int64_t disc = (int64_t)((ha >> 8) % 11);
      const int64_t tax = (int64_t)((ha >> 16) % 9);
      const int64_t part = 1 + (int64_t)((ha >> 24) % n_part);
      const int64_t supp = 1 + (int64_t)((hb >> 40) % n_supp);
      const int32_t ship = odate + 1 + (int32_t)(hb % 121);
      const int32_t commit = odate + 30 + (int32_t)((hb >> 8) % 61);
      const int32_t receipt = ship + 1 + (int32_t)((hb >> 16) % 30);
      const int64_t cents = 90000 + ((part / 10) % 20001) + 100 * (part % 1000);
      if (l_orderkey) l_orderkey[k] = order_key_of_index(j);
      if (l_partkey) l_partkey[k] = part;
      if (l_suppkey) l_suppkey[k] = supp;
      if (l_linenumber) l_linenumber[k] = (int64_t)(o - FIRST_LINE_OF_ORDER[ord]) + 1;
      if (l_quantity) l_quantity[k] = qty;
      if (l_extendedprice) l_extendedprice[k] = (double)(qty * cents) / 100.0;
      if (l_discount) l_discount[k] = (double)disc / 100.0;
      if (l_tax) l_tax[k] = (double)tax / 100.0;
      if (l_shipdate) l_shipdate[k] = ship;
      if (l_commitdate) l_commitdate[k] = commit;
      if (l_receiptdate) l_receiptdate[k] = receipt;
      if (l_returnflag)
        l_returnflag[k] = receipt <= LLKV_TPCH_DATE_1995_06_17 ? (((hb >> 24) & 1) ? 'R' : 'A') : 'N';
      if (l_linestatus) l_linestatus[k] = ship > LLKV_TPCH_DATE_1995_06_17 ? 'O' : 'F';
    }
  });
}

void llkv_tpch_gen_orders(uint64_t seed, double scale, uint64_t row_begin, uint64_t rows,
                          int64_t *o_orderkey, int64_t *o_custkey, int32_t *o_orderdate,
                          int64_t *o_shippriority, int32_t threads) {
  const uint64_t n_cust = llkv_tpch_customers_for_scale(scale);
  parallel_rows(rows, threads, [&](uint64_t b, uint64_t e) {
    for (uint64_t k = b; k < e; ++k) {
      const uint64_t j = row_begin + k;
      const uint64_t h = splitmix64(STREAM_ORDER ^ seed ^ j);
      int64_t cust = 1 + (int64_t)((h >> 16) % n_cust);
      // dbgen never assigns orders to every third customer
      if (cust % 3 == 0) cust = ((uint64_t)cust == n_cust) ? cust - 1 : cust + 1;
      if (cust < 1) cust = 1;
      if (o_orderkey) o_orderkey[k] = order_key_of_index(j);
      if (o_custkey) o_custkey[k] = cust;
      if (o_orderdate) o_orderdate[k] = order_date_of_index(seed, j);
      if (o_shippriority) o_shippriority[k] = 0;
    }
  });
}

void llkv_tpch_gen_customer(uint64_t seed, double /*scale*/, uint64_t row_begin, uint64_t rows,
                            int64_t *c_custkey, uint8_t *c_mktsegment, int32_t threads) {
  parallel_rows(rows, threads, [&](uint64_t b, uint64_t e) {
    for (uint64_t k = b; k < e; ++k) {
      const uint64_t c = row_begin + k;
      if (c_custkey) c_custkey[k] = (int64_t)c + 1;
      if (c_mktsegment) c_mktsegment[k] = (uint8_t)(splitmix64(STREAM_CUSTOMER ^ seed ^ c) % 5);
    }
  });
}

const char *llkv_tpch_segment_name(uint32_t code) {
  static const char *NAMES[5] = {"AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"};
  return code < 5 ? NAMES[code] : "";
}

} // extern "C"
