"""MI355X-native textured Gaussian rasterizer: host side of the C-ABI library libtexgs.so."""
