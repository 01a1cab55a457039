// fdtd3d: zero-argument run of the reference's default problem on the MI355X engine; the problem
// can be resized through FDTD3D_* environment variables (see Sim3DConfig::apply_environment).
#include <exception>
#include <iostream>

#include "FDTD3D.h"

int main() {
    try {
        FDTD3D sim;
        sim.run();
        const auto& c = sim.config();
        std::cout << "[timing] " << c.steps << " steps incl. observables and file output: " << sim.seconds_in_steps() << " s\n";
        std::cout << "[timing] stepping and RMS window on the device: " << sim.seconds_on_device() << " s = "
                  << sim.seconds_on_device() / (c.steps > 0 ? c.steps : 1) * 1e3 << " ms/step\n";
    } catch (const std::exception& e) {
        std::cerr << "fdtd3d: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
