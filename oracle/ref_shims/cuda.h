// TEST INFRASTRUCTURE - host stand-in for <cuda.h> (nothing needed).
#pragma once
