// FieldWriter2D.h -- CSV / dims.txt output in the reference's formats
// (fdtd2d/src/OutputManager.cpp:20-43): "# x,z,val" header, one "x,z,value" line per cell with 16
// significant digits, x outer / z inner; dims.txt with Nx, Nz, dx, dz.  The directory is wiped and
// recreated on construction (OutputManager.cpp:8-18).
#pragma once
#include <string>
#include <vector>

class FieldWriter2D {
public:
    explicit FieldWriter2D(const std::string& directory);
    // values are [Nx][Nz], z fastest
    bool write_csv(const std::string& file_name, const std::vector<double>& values, int Nx, int Nz, double dx, double dz) const;
    bool write_dims(int Nx, int Nz, double dx, double dz) const;

private:
    std::string m_dir;
};
