// Op-stream interpreter over the public NDTFrame / core.h API, for differential testing against the oracle
// (tests/test_host_library.py::test_frame_api_random_operation_sequences, tests/campaigns/host_fuzz_campaign.py).
// The script is a whitespace-separated token stream (numbers as C hex floats, so they arrive bit for bit):
//   frame W H CS OG                    the reference frame (multi-cell), occupancy grid cell OG (0: none)
//   add N x y ...                      NDTFrame::addPoint for each point
//   update tx ty th N x y ...          a one-cell frame holding the N points, then NDTFrame::update(trans, &it)
//   build                              NDTFrame::build, then every created cell: index built mean
//   reset                              NDTFrame::resetCells
//   cost tx ty th N x y ...            cost_function(trans, &frame, &one-cell frame of the N points)
//   pso seed gx gy gth dx dy dth I P N x y ...   srand(seed); pso_optimization(...)
//   align seed gx gy gth N x y ...     srand(seed); NDTFrame::align (default 30 x 50, deviation rule of ndtframe.cpp:253)
//   points                             count and in-order sums of the slot-0 points (cost_function's visiting order)
//   end
// Every result goes to stdout, one line per op, numbers as hex floats.
//   usage: frame_fuzz script.txt
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "ndtpso_slam/core.h"
#include "ndtpso_slam/ndtframe.h"

namespace {

struct Reader {
  FILE* f;
  bool word(char* buf, size_t cap) {
    char fmt[16];
    std::snprintf(fmt, sizeof fmt, "%%%zus", cap - 1);
    return std::fscanf(f, fmt, buf) == 1;
  }
  double num() {
    char buf[64];
    if (!word(buf, sizeof buf)) {
      std::fprintf(stderr, "frame_fuzz: unexpected end of script\n");
      std::exit(2);
    }
    return std::strtod(buf, nullptr);
  }
  long integer() { return (long)num(); }
};

std::unique_ptr<NDTFrame> one_cell_frame(Reader& r, unsigned short w, unsigned short h) {
  // the node's per-scan frame: one cell as large as the frame, no cell parameters (ndtpso_slam_node.cpp:229-230)
  auto f = std::make_unique<NDTFrame>(Vector3d::Zero(), w, h, (double)(w > h ? w : h), false);
  const long n = r.integer();
  for (long i = 0; i < n; ++i) {
    const double x = r.num(), y = r.num();
    Vector2d p(x, y);
    f->addPoint(p);
  }
  return f;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  Reader r{std::fopen(argv[1], "r")};
  if (!r.f) return 2;
  ndtpso_slam_device_init();
  std::unique_ptr<NDTFrame> frame;
  unsigned short W = 0, H = 0;
  char op[32];
  while (r.word(op, sizeof op)) {
    const std::string o(op);
    if (o == "end") break;
    if (o == "frame") {
      W = (unsigned short)r.integer();
      H = (unsigned short)r.integer();
      const double cs = r.num(), og = r.num();
      frame = std::make_unique<NDTFrame>(Vector3d::Zero(), W, H, cs, true, NDTPSOConfig(), og);
      std::printf("frame %u %u\n", (unsigned)frame->widthNumOfCells, (unsigned)frame->heightNumOfCells);
      continue;
    }
    if (!frame) {
      std::fprintf(stderr, "frame_fuzz: '%s' before 'frame'\n", op);
      return 2;
    }
    if (o == "add") {
      const long n = r.integer();
      for (long i = 0; i < n; ++i) {
        const double x = r.num(), y = r.num();
        Vector2d p(x, y);
        frame->addPoint(p);
      }
      std::printf("add %ld\n", n);
    } else if (o == "update") {
      const double tx = r.num(), ty = r.num(), th = r.num();
      auto nf = one_cell_frame(r, W, H);
      frame->update(Vector3d(tx, ty, th), nf.get());
      std::printf("update\n");
    } else if (o == "build") {
      frame->build();
      frame->syncHostView();
      unsigned created = 0;
      for (const NDTCell& c : frame->cells) created += c.created ? 1u : 0u;
      std::printf("build %u", created);
      for (size_t i = 0; i < frame->cells.size(); ++i) {
        const NDTCell& c = frame->cells[i];
        if (c.created) std::printf(" %zu %d %a %a", i, c.built ? 1 : 0, c.mean.x(), c.mean.y());
      }
      std::printf("\n");
    } else if (o == "reset") {
      frame->resetCells();
      std::printf("reset\n");
    } else if (o == "cost") {
      const double tx = r.num(), ty = r.num(), th = r.num();
      auto nf = one_cell_frame(r, W, H);
      std::printf("cost %a\n", cost_function(Vector3d(tx, ty, th), frame.get(), nf.get()));
    } else if (o == "pso") {
      const unsigned seed = (unsigned)r.integer();
      const double gx = r.num(), gy = r.num(), gth = r.num(), dx = r.num(), dy = r.num(), dth = r.num();
      PSOConfig cfg;
      cfg.iterations = (int)r.integer();
      cfg.populationSize = (int)r.integer();
      auto nf = one_cell_frame(r, W, H);
      std::srand(seed);
      const Vector3d pose = pso_optimization(Vector3d(gx, gy, gth), frame.get(), nf.get(), Array3d(dx, dy, dth), cfg);
      std::printf("pso %a %a %a\n", pose.x(), pose.y(), pose.z());
    } else if (o == "align") {
      const unsigned seed = (unsigned)r.integer();
      const double gx = r.num(), gy = r.num(), gth = r.num();
      auto nf = one_cell_frame(r, W, H);
      std::srand(seed);
      const Vector3d pose = frame->align(Vector3d(gx, gy, gth), nf.get());
      std::printf("align %a %a %a\n", pose.x(), pose.y(), pose.z());
    } else if (o == "points") {
      std::vector<double> xy;
      frame->collectPoints(xy);
      double sx = 0., sy = 0.;
      for (size_t i = 0; i + 1 < xy.size(); i += 2) {
        sx += xy[i];
        sy += xy[i + 1];
      }
      std::printf("points %zu %a %a\n", xy.size() / 2, sx, sy);
    } else {
      std::fprintf(stderr, "frame_fuzz: unknown op '%s'\n", op);
      return 2;
    }
  }
  std::fclose(r.f);
  return 0;
}
