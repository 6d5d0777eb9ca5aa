// tests/cpp/mock_dealii/mock_dealii.h — TEST SCAFFOLDING for glue/cracks_gpu_assemble.cc, nothing else.
//
// deal.II, Trilinos and p4est do not exist in this image, so the glue (the deal.II side of the drop-in)
// could never be compiled.  This header declares just the types and members the glue touches -- DoFHandler cell
// iterators, IndexSet, AffineConstraints lines, Epetra_CrsMatrix::ExtractCrsDataPointers / ColMap, TrilinosWrappers block
// vectors as contiguous blocks, 1-rank Utilities::MPI -- with the semantics the glue ASSUMES of the real libraries
// (marked "deal.II-knowledge" there).  It is NOT the reference compiled and it is NOT deal.II: it lets
// tests/cpp/glue_driver.cpp run PfmGlue::rebuild() / assemble() against a mesh the Python tests hand over, so that type
// errors, index logic and the call sequence of the glue are exercised on the GPU and compared with the oracle.
// What it cannot check is whether real deal.II numbers dofs the way the glue assumes; those three assumptions are
// isolated in the glue (global_dof_of, local_column_of_global_dof, the row-map order) with instructions for a real install.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

// ---- MPI (1 rank)
using MPI_Comm = int;
constexpr int MPI_INT = 0, MPI_BYTE = 1;
inline int MPI_Bcast(void *, int, int, int, MPI_Comm) { return 0; }

namespace dealii
{
  namespace types
  {
    using global_dof_index = unsigned int; // deal.II's default build
  }
  struct ExcMessage
  {
    std::string m;
    explicit ExcMessage(std::string s) : m(std::move(s)) {}
  };
#define AssertThrow(cond, exc)                                                                                             \
  do                                                                                                                       \
    {                                                                                                                      \
      if (!(cond))                                                                                                         \
        throw std::runtime_error(std::string("AssertThrow(" #cond "): ") + (exc).m);                                       \
    }                                                                                                                      \
  while (0)

  template <int dim>
  struct Point
  {
    double c[dim] = {};
    Point() = default;
    explicit Point(double x) { c[0] = x; }
    double operator[](unsigned int i) const { return c[i]; }
    double &operator[](unsigned int i) { return c[i]; }
  };
  template <int dim>
  struct GeometryInfo
  {
    static constexpr unsigned int vertices_per_cell = 1u << dim;
  };

  class IndexSet
  {
  public:
    std::vector<types::global_dof_index> idx; // ascending
    bool is_element(types::global_dof_index g) const { return std::binary_search(idx.begin(), idx.end(), g); }
    std::vector<types::global_dof_index>::const_iterator begin() const { return idx.begin(); }
    std::vector<types::global_dof_index>::const_iterator end() const { return idx.end(); }
    std::size_t n_elements() const { return idx.size(); }
  };

  template <int dim>
  struct FiniteElementMock
  {
    unsigned int degree = 1;
    unsigned int n_components() const { return dim + 1; }
  };

  // One active cell: what the glue reads of a DoFCellAccessor
  template <int dim>
  struct CellMock
  {
    bool owned = true, ghost = false;
    types::global_dof_index vdof[1u << dim][dim + 1]; // [vertex][component]
    Point<dim> vert[1u << dim];
    bool is_locally_owned() const { return owned; }
    bool is_ghost() const { return ghost; }
    types::global_dof_index vertex_dof_index(unsigned int v, unsigned int comp) const { return vdof[v][comp]; }
    const Point<dim> &vertex(unsigned int v) const { return vert[v]; }
    Point<dim> center() const
    {
      Point<dim> p;
      for (unsigned int v = 0; v < (1u << dim); ++v)
        for (int d = 0; d < dim; ++d)
          p[d] += vert[v][d] / double(1u << dim);
      return p;
    }
  };
  template <int dim>
  struct CellIteratorMock // `cell->member`, as deal.II's iterators
  {
    const CellMock<dim> *p;
    const CellMock<dim> *operator->() const { return p; }
  };
  template <int dim>
  struct CellRangeMock
  {
    const std::vector<CellMock<dim>> *cells;
    struct It
    {
      const CellMock<dim> *p;
      CellIteratorMock<dim> operator*() const { return CellIteratorMock<dim>{p}; }
      It &operator++()
      {
        ++p;
        return *this;
      }
      bool operator!=(const It &o) const { return p != o.p; }
    };
    It begin() const { return It{cells->data()}; }
    It end() const { return It{cells->data() + cells->size()}; }
  };

  template <int dim>
  class DoFHandler
  {
  public:
    std::vector<CellMock<dim>> cells;
    IndexSet owned, relevant;
    FiniteElementMock<dim> fe;
    types::global_dof_index n_dofs_total = 0;
    CellRangeMock<dim> active_cell_iterators() const { return CellRangeMock<dim>{&cells}; }
    const IndexSet &locally_owned_dofs() const { return owned; }
    const FiniteElementMock<dim> &get_fe() const { return fe; }
    types::global_dof_index n_dofs() const { return n_dofs_total; }
  };
  namespace DoFTools
  {
    template <int dim>
    IndexSet extract_locally_relevant_dofs(const DoFHandler<dim> &dh)
    {
      return dh.relevant;
    }
  } // namespace DoFTools

  template <typename number = double>
  class AffineConstraints
  {
  public:
    using Line = std::vector<std::pair<types::global_dof_index, number>>;
    std::map<types::global_dof_index, Line> lines; // homogeneous lines (cracks.cc:2713, 2878-2879) and hanging nodes
    bool is_constrained(types::global_dof_index g) const { return lines.count(g) != 0; }
    const Line *get_constraint_entries(types::global_dof_index g) const
    {
      const auto it = lines.find(g);
      return it == lines.end() ? nullptr : &it->second;
    }
  };

  namespace Utilities
  {
    namespace MPI
    {
      inline unsigned int this_mpi_process(MPI_Comm) { return 0; }
      inline unsigned int n_mpi_processes(MPI_Comm) { return 1; }
      template <class T>
      T max(const T &x, MPI_Comm) { return x; }
      template <class T>
      T sum(const T &x, MPI_Comm) { return x; }
      template <class T>
      std::vector<T> all_gather(MPI_Comm, const T &x)
      {
        return std::vector<T>(1, x);
      }
      template <class T>
      std::map<unsigned int, T> some_to_some(MPI_Comm, const std::map<unsigned int, T> &req)
      {
        if (!req.empty())
          throw std::runtime_error("mock MPI: a 1-rank run has nobody to ask");
        return {};
      }
    } // namespace MPI
  }   // namespace Utilities
} // namespace dealii

// ---- Epetra: the local CSR of a filled matrix
class Epetra_BlockMap
{
public:
  std::vector<long long> gid; // local id -> global id
  int NumMyElements() const { return (int)gid.size(); }
  long long GID64(int lid) const { return gid[(std::size_t)lid]; }
};
class Epetra_CrsMatrix
{
public:
  std::vector<int> rowptr, colind; // colind: local ids of the column map
  std::vector<double> values;
  Epetra_BlockMap colmap;
  int ExtractCrsDataPointers(int *&rp, int *&ci, double *&v) const
  {
    rp = const_cast<int *>(rowptr.data());
    ci = const_cast<int *>(colind.data());
    v = const_cast<double *>(values.data());
    return 0;
  }
  int NumMyRows() const { return (int)rowptr.size() - 1; }
  const Epetra_BlockMap &ColMap() const { return colmap; }
};

namespace dealii
{
  namespace TrilinosWrappers
  {
    class SparseMatrix
    {
    public:
      Epetra_CrsMatrix A;
      const Epetra_CrsMatrix &trilinos_matrix() const { return A; }
      Epetra_CrsMatrix &trilinos_matrix() { return A; }
    };
    class BlockSparseMatrix
    {
    public:
      SparseMatrix b[2][2];
      SparseMatrix &block(unsigned int r, unsigned int c) { return b[r][c]; }
      const SparseMatrix &block(unsigned int r, unsigned int c) const { return b[r][c]; }
    };
    namespace MPI
    {
      class Vector // the locally owned part, contiguous, in row-map order (ascending global index)
      {
      public:
        std::vector<double> v;
        std::vector<double>::iterator begin() { return v.begin(); }
        std::vector<double>::iterator end() { return v.end(); }
        std::vector<double>::const_iterator begin() const { return v.begin(); }
        std::vector<double>::const_iterator end() const { return v.end(); }
        std::size_t locally_owned_size() const { return v.size(); }
      };
      class BlockVector
      {
      public:
        Vector b[2];
        Vector &block(unsigned int i) { return b[i]; }
        const Vector &block(unsigned int i) const { return b[i]; }
      };
    } // namespace MPI
  }   // namespace TrilinosWrappers
} // namespace dealii
