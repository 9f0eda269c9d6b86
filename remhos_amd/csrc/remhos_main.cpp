// remhos_amd -- command-line front end of the single-GPU driver (rmhd_run), accepting the subset of
// the reference's flags that select the hot path (remhos.cpp:249-334) and printing the same report
// lines (remhos.cpp:1423-1428, 1938-1952):
//   remhos_amd -m periodic-cube -p 10 -rs 4 -o 3 -dt -1 -tf 0.5 -ms 20 -ho 3 -lo 5 -fct 2 -pa [-bt 1 -dtc 1]
#include "../../include/rmh_driver.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

int main(int argc, char **argv)
{
   rmhd_config c;
   std::memset(&c, 0, sizeof(c));
   std::strcpy(c.mesh, "periodic-cube");
   c.rs = 2; c.order = 3; c.problem = 10; c.dt = -1.0; c.t_final = 0.5; c.max_steps = -1;
   c.lo_type = 5; c.fused = 1; c.px = c.py = c.pz = 1; c.rank = 0;
   int ho = 3, fct = 2, device = 0;
   std::string comm_file;
   for (int i = 1; i < argc; i++)
   {
      const std::string a = argv[i];
      auto next = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : ""; };
      if (a == "-m")
      {
         // accept "data/periodic-cube.mesh" as well as the bare lattice name
         std::string m = next();
         const size_t s = m.find_last_of('/');
         if (s != std::string::npos) { m = m.substr(s + 1); }
         const size_t d = m.rfind(".mesh");
         if (d != std::string::npos) { m = m.substr(0, d); }
         std::snprintf(c.mesh, sizeof(c.mesh), "%s", m.c_str());
      }
      else if (a == "-rs") { c.rs = std::atoi(next()); }
      else if (a == "-o") { c.order = std::atoi(next()); }
      else if (a == "-p") { c.problem = std::atoi(next()); }
      else if (a == "-dt") { c.dt = std::atof(next()); }
      else if (a == "-tf") { c.t_final = std::atof(next()); }
      else if (a == "-ms") { c.max_steps = std::atoi(next()); }
      else if (a == "-lo") { c.lo_type = std::atoi(next()); }
      else if (a == "-ho") { ho = std::atoi(next()); }
      else if (a == "-fct") { fct = std::atoi(next()); }
      else if (a == "-bt") { c.bounds_type = std::atoi(next()); }
      else if (a == "-dtc") { c.dt_control = std::atoi(next()); }
      else if (a == "-save") { c.save = 1; }
      else if (a == "-unfused") { c.fused = 0; }
      else if (a == "-px") { c.px = std::atoi(next()); }
      else if (a == "-py") { c.py = std::atoi(next()); }
      else if (a == "-pz") { c.pz = std::atoi(next()); }
      else if (a == "-rank") { c.rank = std::atoi(next()); }
      else if (a == "-dev") { device = std::atoi(next()); }
      else if (a == "-comm-file") { comm_file = next(); }
      else if (a == "-self-wrap") { c.self_wrap = std::atoi(next()); } // validation: 1 + d, see rmh_driver.h
      else if (a == "-warmup") { c.warmup_steps = std::atoi(next()); }
      else if (a == "-pa") { c.pa = 1; }
      else if (a == "-no-vis" || a == "-d") { if (a == "-d") { next(); } }
      else if (a == "-s") { c.ode_solver = std::atoi(next()); }
      else if (a == "-ps") { c.ps = 1; }
      else if (a == "-vb") { c.verify_bounds = 1; } // remhos.cpp:324
      else if (a == "-no-vb") { c.verify_bounds = 0; }
      else if (a == "-tile") { c.tile_rows = std::atoi(next()); } // element numbering of the case builder (rmh_driver.h)
      else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
   }
   if ((ho != 2 && ho != 3) || fct != 2 || c.lo_type < 3 || c.lo_type > 5)
   {
      std::fprintf(stderr, "remhos_amd implements -ho 2|3, -lo 3|4|5, -fct 2, -bt 0|1, -dtc 0|1 (the hot path of SURVEY.md section 8)\n");
      return 1;
   }
   c.ho_type = ho;
   rmhd_result r;
   // box-partitioned runs: all blocks in this process (no -comm-file), or one block per process over RCCL
   //   for r in 0 1; do remhos_amd_run ... -px 2 -rank $r -dev $r -comm-file /tmp/rmh.id & done
   const bool partitioned = c.px * c.py * c.pz > 1 || (c.self_wrap && !comm_file.empty());
   // the one-kernel stage of a partitioned run: rmhd_run_partitioned; the solver classes (-ps, -s 11|12|13, -unfused) of one
   // block per process: rmhd_run_rank
   const bool classes = c.ps || c.ode_solver > 10 || !c.fused;
   int rc;
   if (c.verify_bounds && partitioned && !(classes && !comm_file.empty()))
   {
      std::fprintf(stderr, "remhos_amd: -vb runs with the solver classes (one block, or one block per process with -unfused / -ps / -s 11|12|13)\n");
      return 1;
   }
   if (partitioned && classes && !comm_file.empty()) { rc = rmhd_run_rank(&c, comm_file.c_str(), device, &r, nullptr, nullptr); }
   else if (partitioned) { rc = rmhd_run_partitioned(&c, comm_file.empty() ? nullptr : comm_file.c_str(), device, &r); }
   else { rc = rmhd_run(&c, &r); }
   if (rc != 0)
   {
      std::fprintf(stderr, "remhos_amd: %s\n", rmhd_last_error());
      return 2;
   }
   if (partitioned && c.rank != 0 && !comm_file.empty()) { return 0; } // (rank 0 reports, remhos.cpp:1423)
   std::printf("Number of unknowns: %lld\n", r.global_dofs);
   std::printf("time step: %d, time: %.8g, dt: %.8g\n", r.steps, r.t_end, r.dt);
   if (c.dt_control) { std::printf("Total time steps: %d (%d repeated).\n", r.steps + r.repeats, r.repeats); } // remhos.cpp:1350-1354
   std::printf("---\nRHS   kernel time: %.8g\nL2inv kernel time: %.8g\nLO    kernel time: %.8g\nFCT   kernel time: %.8g\n"
               "Total kernel time: %.8g\n---\n", r.t_rhs, r.t_inv, r.t_lo, r.t_fct, r.t_total);
   std::printf("FOM RHS: %.8g\nFOM INV: %.8g\nFOM LO:  %.8g\nFOM FCT: %.8g\nFOM:     %.8g\n"
               "(megadofs x time steps / second)\n---\n", r.fom_rhs, r.fom_inv, r.fom_lo, r.fom_fct, r.fom);
   if (r.timer_every > 0) { std::printf("(kernel times and FOMs above: sampled on %d of the timed steps, every %d-th, and scaled; the wall-clock FOM below is exact)\n", r.timer_steps, r.timer_every); }
   std::printf("FOM wall (everything included): %.8g\nmax local PCG iterations: %d\n", r.fom_wall, r.cg_iters_max);
   std::printf("Final mass u:  %.10g\nMax value u:   %.10g\nMass loss u:   %.6g\n", r.final_mass, r.max_value, r.mass_loss);
   if (r.has_errors) { std::printf("L1-error: %.10g. (L2 %.10g, Linf %.10g)\n", r.err_l1, r.err_l2, r.err_linf); } // remhos.cpp:1441-1442
   if (c.ps) // remhos.cpp:1429-1435
   {
      std::printf("Final mass us: %.10g\nMax value s:   %.10g\nMass loss us:  %.6g\n", r.final_mass_us, r.s_max, r.mass_loss_us);
   }
   return 0;
}
