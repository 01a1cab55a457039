// FieldWriter3D.h -- writes cell-centred scalars as ASCII VTK ImageData (.vti) in exactly the
// layout the reference's writer produces (fdtd3d/src/OutputManager.cpp:23-84): Float64 values with
// 16 significant digits, four per line, header with WholeExtent/Origin/Spacing; the directory is
// wiped and recreated on construction (OutputManager.cpp:9-21).
#pragma once
#include <string>
#include <vector>

class FieldWriter3D {
public:
    // ascii: the reference's format, byte for byte.  appended: same header and data array, but the
    // values follow as raw little-endian Float32 in an <AppendedData encoding="raw"> block (UInt64
    // length prefix) -- 4 bytes per cell instead of ~20, for grids where ASCII is not an option.
    enum class Encoding { ascii, appended };
    explicit FieldWriter3D(const std::string& directory, Encoding encoding = Encoding::ascii);
    // values are [Nz][Ny][Nx], x fastest; file <dir>/<name>_t<step, 5 digits>.vti
    bool write_vti(const std::string& name, const std::vector<double>& values, int Nx, int Ny, int Nz,
                   double dx, double dy, double dz, int step) const;
    const std::string& directory() const { return m_dir; }

private:
    std::string m_dir;
    Encoding m_encoding;
};
