// tests/cpp/glue_driver.cpp — TEST SCAFFOLDING: runs glue/cracks_gpu_assemble.cc (the deal.II side of the drop-in,
// SURVEY.md 8(f) N4) against tests/cpp/mock_dealii/ on a problem the Python test writes into a directory, and writes
// what the glue left in the "Trilinos" objects back for comparison with the oracle (tests/test_glue_mock.py).
// This is NOT deal.II and NOT the reference: see tests/cpp/mock_dealii/mock_dealii.h.
//
//   glue_driver <dir>     reads <dir>/meta.txt + *.bin, calls PfmGlue::rebuild() once, PfmGlue::assemble(residual_only)
//                         for both modes, writes <dir>/out_*.bin
#define PFM_WITH_DEALII
#include "mock_dealii.h"

#include "../../glue/cracks_gpu_assemble.cc"

#include <cstdio>
#include <cmath>
#include <fstream>
#include <sstream>

using namespace dealii;
using gidx = types::global_dof_index;

template <class T>
static std::vector<T> read_bin(const std::string &path)
{
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f)
    throw std::runtime_error("cannot open " + path);
  const std::streamsize n = f.tellg();
  f.seekg(0);
  std::vector<T> v((size_t)n / sizeof(T));
  f.read(reinterpret_cast<char *>(v.data()), n);
  return v;
}
template <class T>
static void write_bin(const std::string &path, const T *p, size_t n)
{
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char *>(p), (std::streamsize)(n * sizeof(T)));
}

struct ConstFunction1
{
  double v = 0.0;
  double value(const Point<1> &, unsigned int) const { return v; }
};
// a smooth stand-in for the reference's bitmap function (cracks.cc:118-241): the test evaluates the same formula at the
// cell centres for the oracle (tests/test_glue_mock.py: emodulus_mock)
template <int dim>
struct EModulusMock
{
  double value(const Point<dim> &p, unsigned int) const
  {
    double e = 100.0 + 50.0 * std::sin(0.7 * p[0]) * std::cos(0.3 * p[1]);
    if (dim == 3)
      e += 20.0 * std::sin(0.5 * p[dim - 1]);
    return e;
  }
};

// the members of FracturePhaseFieldProblem<dim> the glue touches (cracks.cc:1039-1180)
template <int dim>
struct Problem
{
  enum class TestCase
  {
    sneddon,
    multiple_het
  };
  enum class OuterSolverType
  {
    active_set,
    simple_monolithic
  };
  DoFHandler<dim> dof_handler;
  MPI_Comm mpi_com = 0;
  bool direct_solver = false;
  TestCase test_case = TestCase::sneddon;
  OuterSolverType outer_solver = OuterSolverType::active_set;
  EModulusMock<dim> emod, *func_emodulus = &emod;
  ConstFunction1 func_pressure;
  double poisson_ratio_nu = 0.2, lame_coefficient_lambda = 0, lame_coefficient_mu = 0, G_c = 0, alpha_eps = 0, constant_k = 0, time = 0,
         alpha_biot = 0, gamma_penal = 0, timestep = 1, old_timestep = 1, old_old_timestep = 1, decompose_stress_rhs = 0,
         decompose_stress_matrix = 0;
  unsigned int timestep_number = 0;
  bool use_old_timestep_pf = false;
  AffineConstraints<double> constraints_hanging_nodes, constraints_update;
  TrilinosWrappers::BlockSparseMatrix system_pde_matrix;
  TrilinosWrappers::MPI::BlockVector solution, old_solution, old_old_solution, system_pde_residual, system_total_residual;
  // the two helpers the glue asks of the problem class
  gidx n_u_dofs_global() const { return dof_handler.n_dofs() / (dim + 1) * dim; }
  bool is_phase_field_dof(gidx g) const { return direct_solver ? (g % (dim + 1) == dim) : (g >= n_u_dofs_global()); }
};

template <int dim>
static int run(const std::string &dir, std::istringstream &meta)
{
  constexpr int nv = 1 << dim;
  int blocked, n_nodes, n_cells, n_hanging, solver;
  meta >> blocked >> n_nodes >> n_cells >> n_hanging >> solver;
  Problem<dim> P;
  P.direct_solver = !blocked;
  P.outer_solver = solver == 0 ? Problem<dim>::OuterSolverType::active_set : Problem<dim>::OuterSolverType::simple_monolithic;
  double pressure;
  int use_old, tsn;
  meta >> P.lame_coefficient_lambda >> P.lame_coefficient_mu >> P.G_c >> P.alpha_eps >> P.constant_k >> pressure >> P.alpha_biot >> P.gamma_penal >>
    P.timestep >> P.time >> P.old_timestep >> P.old_old_timestep >> P.decompose_stress_rhs >> P.decompose_stress_matrix >> tsn >> use_old;
  int het = 0;
  if (meta >> het >> P.poisson_ratio_nu) // optional: a multiple_het run (cracks.cc:2207-2216) with this Poisson ratio
    P.test_case = het ? Problem<dim>::TestCase::multiple_het : Problem<dim>::TestCase::sneddon;
  P.func_pressure.v = pressure;
  P.timestep_number = (unsigned int)tsn;
  P.use_old_timestep_pf = use_old != 0;

  // the numbering the glue ASSUMES of deal.II (deal.II-knowledge, see its header): vertex rank r ->
  //   interleaved: dofs (dim + 1) r + comp;   component_wise blocks {u, phi}: u dofs dim r + comp, phi dof n_u + r
  const gidx n_u = (gidx)dim * n_nodes;
  auto dof_of = [&](int r, int comp) -> gidx {
    if (!blocked)
      return (gidx)((dim + 1) * r + comp);
    return comp < dim ? (gidx)(dim * r + comp) : n_u + (gidx)r;
  };
  const auto cells = read_bin<int32_t>(dir + "/cells.bin");
  const auto coords = read_bin<double>(dir + "/coords.bin");
  P.dof_handler.n_dofs_total = (gidx)(dim + 1) * n_nodes;
  P.dof_handler.cells.resize((size_t)n_cells);
  for (int c = 0; c < n_cells; ++c)
    for (int vtx = 0; vtx < nv; ++vtx)
      {
        const int r = cells[(size_t)c * nv + vtx];
        for (int comp = 0; comp <= dim; ++comp)
          P.dof_handler.cells[c].vdof[vtx][comp] = dof_of(r, comp);
        for (int d = 0; d < dim; ++d)
          P.dof_handler.cells[c].vert[vtx][d] = coords[(size_t)r * dim + d];
      }
  for (gidx g = 0; g < P.dof_handler.n_dofs_total; ++g)
    P.dof_handler.owned.idx.push_back(g);
  P.dof_handler.relevant = P.dof_handler.owned;
  // hanging nodes: one line per component (cracks.cc:1630-1635), then the homogeneous lines of constraints_update
  if (n_hanging > 0)
    {
      const auto hn = read_bin<int32_t>(dir + "/hn_nodes.bin"), hp = read_bin<int32_t>(dir + "/hn_parents.bin");
      const auto ptr = read_bin<int64_t>(dir + "/hn_ptr.bin");
      const auto w = read_bin<double>(dir + "/hn_w.bin");
      for (int k = 0; k < n_hanging; ++k)
        for (int comp = 0; comp <= dim; ++comp)
          {
            AffineConstraints<double>::Line line;
            for (int64_t e = ptr[k]; e < ptr[k + 1]; ++e)
              line.emplace_back(dof_of(hp[e], comp), w[e]);
            P.constraints_hanging_nodes.lines[dof_of(hn[k], comp)] = line;
          }
    }
  P.constraints_update = P.constraints_hanging_nodes;
  {
    const auto fl = read_bin<uint8_t>(dir + "/con_update.bin");
    for (int r = 0; r < n_nodes; ++r)
      for (int comp = 0; comp <= dim; ++comp)
        if ((fl[r] >> comp) & 1u)
          P.constraints_update.lines[dof_of(r, comp)] = {};
  }
  // vectors: owned part per block, contiguous, ascending global index
  auto fill = [&](TrilinosWrappers::MPI::BlockVector &v, const std::vector<double> &x) {
    if (blocked)
      {
        v.block(0).v.assign(x.begin(), x.begin() + n_u);
        v.block(1).v.assign(x.begin() + n_u, x.end());
      }
    else
      v.block(0).v = x;
  };
  fill(P.solution, read_bin<double>(dir + "/sol.bin"));
  fill(P.old_solution, read_bin<double>(dir + "/old.bin"));
  fill(P.old_old_solution, read_bin<double>(dir + "/oldold.bin"));
  const std::vector<double> zeros((size_t)(dim + 1) * n_nodes, 0.0);
  fill(P.system_pde_residual, zeros);
  fill(P.system_total_residual, zeros);
  // matrices: Epetra's local CSR per block; the column map is the identity here (1 rank), colind = block-local global ids
  const int nb1 = blocked ? 2 : 1;
  for (int r = 0; r < nb1; ++r)
    for (int c = 0; c < nb1; ++c)
      {
        Epetra_CrsMatrix &A = P.system_pde_matrix.block(r, c).trilinos_matrix();
        const std::string tag = std::to_string(2 * r + c);
        A.rowptr = read_bin<int>(dir + "/rowptr" + tag + ".bin");
        A.colind = read_bin<int>(dir + "/colind" + tag + ".bin");
        A.values.assign(A.colind.size(), -7.0e77); // every value must be overwritten
        const long long ncol = blocked ? (c == 0 ? (long long)n_u : (long long)n_nodes) : (long long)(dim + 1) * n_nodes;
        A.colmap.gid.resize((size_t)ncol);
        for (long long g = 0; g < ncol; ++g)
          A.colmap.gid[(size_t)g] = g;
      }

  pfm_glue_detail::PfmGlue<dim> glue;
  glue.pin_host_matrix = true;
  glue.rebuild(P);
  {
    // a second setup_system(): without before_setup_system() the glue must refuse (the locks of the first rebuild are on
    // arrays the host is about to free), with it the rebuild goes through
    bool refused = false;
    try
      {
        glue.rebuild(P);
      }
    catch (const std::exception &)
      {
        refused = true;
      }
    if (!refused)
      throw std::runtime_error("rebuild() over a context that holds page locks was not refused");
    glue.before_setup_system();
    glue.rebuild(P);
  }
  auto dump = [&](const TrilinosWrappers::MPI::BlockVector &v, const std::string &name) {
    std::vector<double> x;
    for (int b = 0; b < nb1; ++b)
      x.insert(x.end(), v.block(b).begin(), v.block(b).end());
    write_bin(dir + "/" + name, x.data(), x.size());
  };
  glue.assemble(P, /*residual_only=*/true);
  glue.assemble(P, /*residual_only=*/true, /*only_solution_changed=*/true); // the line-search form: same state, same result
  dump(P.system_pde_residual, "out_res_pde_ro.bin");
  dump(P.system_total_residual, "out_res_tot_ro.bin");
  {
    // the line search without the vectors (residual_to_host = false): the norms come from the device copy
    glue.residual_to_host = false;
    glue.assemble(P, /*residual_only=*/true, /*only_solution_changed=*/true);
    const double n_pde = glue.residual_l2_norm(P), n_tot = glue.residual_l2_norm(P, true);
    glue.residual_to_host = true;
    write_bin(dir + "/out_norms.bin", std::vector<double>{n_pde, n_tot}.data(), (size_t)2);
  }
  glue.assemble(P, /*residual_only=*/false);
  dump(P.system_pde_residual, "out_res_pde.bin");
  for (int r = 0; r < nb1; ++r)
    for (int c = 0; c < nb1; ++c)
      {
        const Epetra_CrsMatrix &A = P.system_pde_matrix.block(r, c).trilinos_matrix();
        write_bin(dir + "/out_val" + std::to_string(2 * r + c) + ".bin", A.values.data(), A.values.size());
      }
  int path = pfm_ctx_kernel_path(glue.ctx);
  std::printf("glue_driver: OK (dim %d, %s layout, %d nodes, %d cells, %d hanging, kernel path %d)\n", dim, blocked ? "blocked" : "interleaved",
              n_nodes, n_cells, n_hanging, path);
  return 0;
}

int main(int argc, char **argv)
{
  if (argc < 2)
    {
      std::fprintf(stderr, "usage: glue_driver <dir>\n");
      return 2;
    }
  try
    {
      const std::string dir = argv[1];
      std::ifstream f(dir + "/meta.txt");
      std::stringstream ss;
      ss << f.rdbuf();
      std::istringstream meta(ss.str());
      int dim;
      meta >> dim;
      return dim == 2 ? run<2>(dir, meta) : run<3>(dir, meta);
    }
  catch (const std::exception &e)
    {
      std::fprintf(stderr, "glue_driver: FAILED: %s\n", e.what());
      return 1;
    }
}
