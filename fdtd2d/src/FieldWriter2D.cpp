#include "FieldWriter2D.h"

#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>

namespace fs = std::filesystem;

FieldWriter2D::FieldWriter2D(const std::string& directory) : m_dir(fs::absolute(directory).string()) {
    std::error_code ec;
    fs::remove_all(m_dir, ec);
    fs::create_directory(m_dir, ec);
    if (ec) std::cerr << "[FieldWriter2D] cannot prepare " << m_dir << ": " << ec.message() << std::endl;
}

bool FieldWriter2D::write_csv(const std::string& file_name, const std::vector<double>& values, int Nx, int Nz, double dx, double dz) const {
    std::ofstream out(m_dir + "/" + file_name);
    if (!out) return false;
    out << "# x,z,val\n" << std::setprecision(16);
    for (int i = 0; i < Nx; ++i)
        for (int k = 0; k < Nz; ++k)
            out << (i * dx) << "," << (k * dz) << "," << values[static_cast<size_t>(i) * Nz + k] << "\n";
    return true;
}

bool FieldWriter2D::write_dims(int Nx, int Nz, double dx, double dz) const {
    std::ofstream out(m_dir + "/dims.txt");
    if (!out) return false;
    out << "Nx " << Nx << "\n" << "Nz " << Nz << "\n" << std::setprecision(16) << "dx " << dx << "\n" << "dz " << dz << "\n";
    return true;
}
