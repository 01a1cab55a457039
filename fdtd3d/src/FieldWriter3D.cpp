#include "FieldWriter3D.h"

#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>

namespace fs = std::filesystem;

FieldWriter3D::FieldWriter3D(const std::string& directory, Encoding encoding) : m_dir(fs::absolute(directory).string()), m_encoding(encoding) {
    std::error_code ec;
    fs::remove_all(m_dir, ec);
    fs::create_directory(m_dir, ec);
    if (ec) std::cerr << "[FieldWriter3D] cannot prepare " << m_dir << ": " << ec.message() << std::endl;
}

bool FieldWriter3D::write_vti(const std::string& name, const std::vector<double>& values, int Nx, int Ny, int Nz,
                              double dx, double dy, double dz, int step) const {
    std::ostringstream path;
    path << m_dir << "/" << name << "_t" << std::setw(5) << std::setfill('0') << step << ".vti";
    std::ofstream out(path.str(), m_encoding == Encoding::appended ? std::ios::binary : std::ios::out);
    if (!out) {
        std::cerr << "[FieldWriter3D] cannot open " << path.str() << std::endl;
        return false;
    }
    std::ostringstream extent;
    extent << 0 << " " << Nx - 1 << " " << 0 << " " << Ny - 1 << " " << 0 << " " << Nz - 1;
    if (m_encoding == Encoding::appended) {
        out << "<?xml version=\"1.0\"?>\n"
            << "<VTKFile type=\"ImageData\" version=\"1.0\" byte_order=\"LittleEndian\" header_type=\"UInt64\">\n"
            << "  <ImageData WholeExtent=\"" << extent.str() << "\" Origin=\"0 0 0\" Spacing=\"" << dx << " " << dy << " " << dz << "\">\n"
            << "    <Piece Extent=\"" << extent.str() << "\">\n"
            << "      <PointData Scalars=\"" << name << "\">\n"
            << "        <DataArray type=\"Float32\" Name=\"" << name << "\" format=\"appended\" offset=\"0\" NumberOfComponents=\"1\"/>\n"
            << "      </PointData>\n    </Piece>\n  </ImageData>\n  <AppendedData encoding=\"raw\">\n   _";
        std::vector<float> f(values.begin(), values.end());
        const unsigned long long nbytes = f.size() * sizeof(float);
        out.write(reinterpret_cast<const char*>(&nbytes), sizeof nbytes);
        out.write(reinterpret_cast<const char*>(f.data()), static_cast<std::streamsize>(nbytes));
        out << "\n  </AppendedData>\n</VTKFile>\n";
        out.close();
        std::cout << "[OutputManager] Saved VTI: " << path.str() << std::endl;
        return static_cast<bool>(out);
    }
    out << "<?xml version=\"1.0\"?>\n"
        << "<VTKFile type=\"ImageData\" version=\"1.0\" byte_order=\"LittleEndian\">\n"
        << "  <ImageData WholeExtent=\"" << extent.str() << "\" Origin=\"0 0 0\" Spacing=\"" << dx << " " << dy << " " << dz << "\">\n"
        << "    <Piece Extent=\"" << extent.str() << "\">\n"
        << "      <PointData Scalars=\"" << name << "\">\n"
        << "        <DataArray type=\"Float64\" Name=\"" << name << "\" format=\"ascii\" NumberOfComponents=\"1\">\n";
    out << std::setprecision(16);
    const size_t n = values.size();
    for (size_t q = 0; q < n; ++q) {
        if (q > 0 && q % 4 == 0) out << "\n";
        out << values[q] << " ";
    }
    if (n) out << "\n";
    out << "        </DataArray>\n      </PointData>\n    </Piece>\n  </ImageData>\n</VTKFile>\n";
    out.close();
    std::cout << "[OutputManager] Saved VTI: " << path.str() << std::endl;
    return true;
}
