#define RT_KERNEL_SOURCE_HASH "cfec2feda8d720cf"
