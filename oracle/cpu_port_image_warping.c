/* placeholder translation unit; filled in below */
int orc_cpu_port_placeholder(void) { return 0; }
