// fdtd2d: zero-argument run of the reference's default problem on the MI355X engine; resize through
// FDTD2D_* environment variables (see Sim2DConfig::apply_environment).
#include <exception>
#include <iostream>

#include "FDTD2D.h"

int main() {
    try {
        FDTD2D sim;
        sim.run();
        std::cout << "[timing] " << sim.config().steps << " steps incl. observables and file output: " << sim.seconds_in_steps() << " s\n";
        std::cout << "[timing] stepping and RMS window on the device: " << sim.seconds_on_device() << " s = "
                  << sim.seconds_on_device() / sim.config().steps * 1e3 << " ms/step\n";
    } catch (const std::exception& e) {
        std::cerr << "fdtd2d: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
